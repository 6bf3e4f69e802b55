// libugrid_hip.so -- drop-in kernels behind the reference's four extension modules
// (render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda), written for
// gfx950 (MI355X): 64-lane waves, one wave per ray for the scans, 16-byte vector streams for
// the per-voxel optimiser/TV passes.  fp32 arithmetic follows the reference expression trees
// (compiled with -ffp-contract=off) so results are bit-identical to oracle/ref_ops.c except
// where libm transcendentals (exp/pow) are involved.
//
// Reference behaviour restated (never copied): FourierGrid/cuda/render_utils_kernel.cu,
// adam_upd_kernel.cu, total_variation_kernel.cu, ub360_utils_kernel.cu -- per-function
// file:line citations are in include/ugrid_hip.h.
#include "ugrid_common.h"

extern "C" int ugrid_abi_version(void) { return 2; }  // 2: ugrid_render_params.mlp_mode, ugrid_pack_mlp(k0_absmax, best_mode)
extern "C" const char *ugrid_target_arch(void) { return "gfx950"; }

// ----------------------------------------------------------------------------------------------
// Ray / AABB helpers (1 lane per ray; 12-byte AoS rays are read as 3 dwords, L1 absorbs the stride)
// ----------------------------------------------------------------------------------------------
struct ug_tmm { float tmin, tmax; };

__device__ __forceinline__ ug_tmm ug_t_minmax(const float *o, const float *d, const float *lo,
                                              const float *hi, float near, float far) {
  // a zero direction component is replaced by float(1e-6) (double literal narrowed)
  const float vx = (d[0] == 0.f) ? (float)1e-6 : d[0];
  const float vy = (d[1] == 0.f) ? (float)1e-6 : d[1];
  const float vz = (d[2] == 0.f) ? (float)1e-6 : d[2];
  const float ax = (hi[0] - o[0]) / vx, ay = (hi[1] - o[1]) / vy, az = (hi[2] - o[2]) / vz;
  const float bx = (lo[0] - o[0]) / vx, by = (lo[1] - o[1]) / vy, bz = (lo[2] - o[2]) / vz;
  ug_tmm r;
  r.tmin = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), far), near);
  r.tmax = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), far), near);
  return r;
}

__device__ __forceinline__ float ug_norm3(const float *d) {
  return sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

__device__ __forceinline__ int64_t ug_n_samples(const float *d, float tmin, float tmax, float stepdist) {
  const double c = (double)ceilf((tmax - tmin) * ug_norm3(d) / stepdist);
  return (int64_t)(c > 1. ? c : 1.);
}

__global__ void k_infer_t_minmax(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                 const float *__restrict__ xyz_min, const float *__restrict__ xyz_max,
                                 float near, float far, int64_t n_rays, float *__restrict__ t_min,
                                 float *__restrict__ t_max) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float lo[3] = {xyz_min[0], xyz_min[1], xyz_min[2]}, hi[3] = {xyz_max[0], xyz_max[1], xyz_max[2]};
  const ug_tmm t = ug_t_minmax(rays_o + 3 * r, rays_d + 3 * r, lo, hi, near, far);
  t_min[r] = t.tmin;
  t_max[r] = t.tmax;
}

__global__ void k_infer_n_samples(const float *__restrict__ rays_d, const float *__restrict__ t_min,
                                  const float *__restrict__ t_max, float stepdist, int64_t n_rays,
                                  int64_t *__restrict__ n_samples) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  n_samples[r] = ug_n_samples(rays_d + 3 * r, t_min[r], t_max[r], stepdist);
}

__global__ void k_infer_ray_start_dir(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                      const float *__restrict__ t_min, int64_t n_rays,
                                      float *__restrict__ rays_start, float *__restrict__ rays_dir) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
  const float rn = ug_norm3(d), tm = t_min[r];
  for (int c = 0; c < 3; ++c) {
    rays_start[3 * r + c] = o[c] + d[c] * tm;
    rays_dir[3 * r + c] = d[c] / rn;
  }
}

// fused first half of sample_pts_on_rays: t_min, t_max, N_steps in one pass over the rays
__global__ void k_sample_count(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                               const float *__restrict__ xyz_min, const float *__restrict__ xyz_max,
                               float near, float far, float stepdist, int64_t n_rays,
                               float *__restrict__ t_min, float *__restrict__ t_max,
                               int64_t *__restrict__ n_steps) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rays) return;
  const float lo[3] = {xyz_min[0], xyz_min[1], xyz_min[2]}, hi[3] = {xyz_max[0], xyz_max[1], xyz_max[2]};
  const ug_tmm t = ug_t_minmax(rays_o + 3 * r, rays_d + 3 * r, lo, hi, near, far);
  t_min[r] = t.tmin;
  t_max[r] = t.tmax;
  n_steps[r] = ug_n_samples(rays_d + 3 * r, t.tmin, t.tmax, stepdist);
}

// ----------------------------------------------------------------------------------------------
// int64 inclusive scan (three short kernels; the ray counts involved are <= a few million)
// ----------------------------------------------------------------------------------------------
#define UG_SCAN_THREADS 256
#define UG_SCAN_ITEMS 4
#define UG_SCAN_TILE (UG_SCAN_THREADS * UG_SCAN_ITEMS)

__device__ __forceinline__ int64_t ug_block_exclusive_scan(int64_t v, int64_t *lds, int64_t *block_total) {
  // Hillis-Steele over 256 per-thread sums held in LDS
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int off = 1; off < UG_SCAN_THREADS; off <<= 1) {
    const int64_t add = (t >= off) ? lds[t - off] : 0;
    __syncthreads();
    lds[t] += add;
    __syncthreads();
  }
  *block_total = lds[UG_SCAN_THREADS - 1];
  return lds[t] - v;
}

__global__ void __launch_bounds__(UG_SCAN_THREADS)
k_scan_local(const int64_t *__restrict__ in, int64_t n, int64_t *__restrict__ out,
             int64_t *__restrict__ block_sums) {
  __shared__ int64_t lds[UG_SCAN_THREADS];
  const int64_t base = (int64_t)blockIdx.x * UG_SCAN_TILE + (int64_t)threadIdx.x * UG_SCAN_ITEMS;
  int64_t v[UG_SCAN_ITEMS], s = 0;
  for (int i = 0; i < UG_SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  int64_t total;
  int64_t run = ug_block_exclusive_scan(s, lds, &total);
  for (int i = 0; i < UG_SCAN_ITEMS; ++i) {
    run += v[i];
    if (base + i < n) out[base + i] = run;
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(UG_SCAN_THREADS)
k_scan_block_sums(int64_t *__restrict__ block_sums, int64_t n_blocks, int64_t *__restrict__ total_out) {
  __shared__ int64_t lds[UG_SCAN_THREADS];
  int64_t carry = 0;
  for (int64_t base = 0; base < n_blocks; base += UG_SCAN_THREADS) {
    const int64_t i = base + threadIdx.x;
    const int64_t v = (i < n_blocks) ? block_sums[i] : 0;
    int64_t total;
    const int64_t ex = ug_block_exclusive_scan(v, lds, &total);
    if (i < n_blocks) block_sums[i] = carry + ex;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ void k_scan_add(int64_t *__restrict__ out, int64_t n, const int64_t *__restrict__ block_sums) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += block_sums[i / UG_SCAN_TILE];
}

extern "C" int64_t ugrid_scan_ws_bytes(int64_t n) {
  return (int64_t)sizeof(int64_t) * ((n + UG_SCAN_TILE - 1) / UG_SCAN_TILE + 1);
}

static int ug_inclusive_scan(const int64_t *in, int64_t n, int64_t *out, int64_t *d_total, void *ws,
                             hipStream_t st) {
  if (n == 0) return (int)hipMemsetAsync(d_total, 0, sizeof(int64_t), st);
  const int64_t nb = (n + UG_SCAN_TILE - 1) / UG_SCAN_TILE;
  int64_t *bs = (int64_t *)ws;
  hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nb), dim3(UG_SCAN_THREADS), 0, st, in, n, out, bs);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(UG_SCAN_THREADS), 0, st, bs, nb, d_total);
  hipLaunchKernelGGL(k_scan_add, dim3(ug_blocks(n, 256)), dim3(256), 0, st, out, n, bs);
  UG_LAUNCH_CHECK();
  return 0;
}

// second half of sample_pts_on_rays: 1 lane per sample, owner ray by binary search in the
// inclusive prefix sum (replaces the reference's "1 at segment start + cumsum" construction).
__global__ void k_sample_fill(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                              const float *__restrict__ xyz_min, const float *__restrict__ xyz_max,
                              const float *__restrict__ t_min, const int64_t *__restrict__ cumsum,
                              float stepdist, int64_t n_rays, int64_t total,
                              float *__restrict__ rays_pts, uint8_t *__restrict__ mask_outbbox,
                              int64_t *__restrict__ ray_id, int64_t *__restrict__ step_id) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int64_t lo = 0, hi = n_rays - 1;  // first r with cumsum[r] > idx
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (cumsum[mid] > idx) hi = mid; else lo = mid + 1;
  }
  const int64_t r = lo;
  const int64_t s = idx - (r ? cumsum[r - 1] : 0);
  const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
  const float rn = ug_norm3(d), tm = t_min[r];
  const float dist = stepdist * (float)(int)s;
  float p[3];
  for (int c = 0; c < 3; ++c) {
    const float start = o[c] + d[c] * tm;
    const float dir = d[c] / rn;
    p[c] = start + dir * dist;
    rays_pts[3 * idx + c] = p[c];
  }
  mask_outbbox[idx] = (uint8_t)((xyz_min[0] > p[0]) | (xyz_min[1] > p[1]) | (xyz_min[2] > p[2]) |
                                (xyz_max[0] < p[0]) | (xyz_max[1] < p[1]) | (xyz_max[2] < p[2]));
  ray_id[idx] = r;
  step_id[idx] = s;
}

__global__ void k_sample_ndc(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                             const float *__restrict__ xyz_min, const float *__restrict__ xyz_max,
                             int n_samples, int64_t total, float *__restrict__ rays_pts,
                             uint8_t *__restrict__ mask_outbbox) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t r = idx / n_samples;
  const int s = (int)(idx - r * n_samples);
  const float dist = ((float)s) / (float)(n_samples - 1);
  float p[3];
  for (int c = 0; c < 3; ++c) {
    p[c] = rays_o[3 * r + c] + rays_d[3 * r + c] * dist;
    rays_pts[3 * idx + c] = p[c];
  }
  mask_outbbox[idx] = (uint8_t)((xyz_min[0] > p[0]) | (xyz_min[1] > p[1]) | (xyz_min[2] > p[2]) |
                                (xyz_max[0] < p[0]) | (xyz_max[1] < p[1]) | (xyz_max[2] < p[2]));
}

__global__ void k_sample_bg(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                            const float *__restrict__ t_max, float bg_preserve, int n_samples,
                            int64_t total, float *__restrict__ rays_pts) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t r = idx / n_samples;
  const int s = (int)(idx - r * n_samples);
  const float frac = ((float)s) / (float)n_samples;
  const float t_out = (float)((double)t_max[r] - 1. + 1. / (1. - (double)frac));
  const float x = rays_o[3 * r] + rays_d[3 * r] * t_out;
  const float y = rays_o[3 * r + 1] + rays_d[3 * r + 1] * t_out;
  const float z = rays_o[3 * r + 2] + rays_d[3 * r + 2] * t_out;
  const float tn = sqrtf(x * x + y * y + z * z);
  const float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
  const float Ro = tn / m;
  const float q = (float)((double)(Ro * Ro / (tn * tn)) * (1. - (double)bg_preserve) +
                          (double)(Ro / tn * bg_preserve));
  rays_pts[3 * idx] = x * q;
  rays_pts[3 * idx + 1] = y * q;
  rays_pts[3 * idx + 2] = z * q;
}

__global__ void k_maskcache(const uint8_t *__restrict__ world, const float *__restrict__ xyz,
                            const float *__restrict__ scale, const float *__restrict__ shift,
                            int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n,
                            uint8_t *__restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float fi = roundf(xyz[3 * p] * scale[0] + shift[0]);
  float fj = roundf(xyz[3 * p + 1] * scale[1] + shift[1]);
  float fk = roundf(xyz[3 * p + 2] * scale[2] + shift[2]);
  // the reference converts the rounded value with `const int i = round(...)` (render_utils_kernel.cu:385-387): the
  // hardware float->int conversion saturates and maps NaN to 0 (v_cvt_i32_f32, and cvt.rzi.s32.f32 on the
  // reference's own target), so a NaN coordinate indexes plane 0 of that axis -- pinned on the reference kernels
  // themselves (tests/golden/native_ops.npz); +-inf / huge values saturate out of range
  fi = (fi != fi) ? 0.f : fi; fj = (fj != fj) ? 0.f : fj; fk = (fk != fk) ? 0.f : fk;
  uint8_t v = 0;
  if (fi >= 0.f && fi < (float)sz_i && fj >= 0.f && fj < (float)sz_j && fk >= 0.f && fk < (float)sz_k)
    v = world[(int64_t)fi * sz_j * sz_k + (int64_t)fj * sz_k + (int64_t)fk];
  out[p] = v;
}

// ----------------------------------------------------------------------------------------------
// raw -> alpha  (12 B/point stream)
// ----------------------------------------------------------------------------------------------
__global__ void k_raw2alpha(const float *__restrict__ density, float shift, float interval,
                            const float *__restrict__ interval_arr, int64_t n,
                            float *__restrict__ exp_d, float *__restrict__ alpha) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float itv = interval_arr ? interval_arr[i] : interval;
  const float e = expf(density[i] + shift);
  exp_d[i] = e;
  alpha[i] = 1 - powf(1 + e, -itv);
}

__global__ void k_raw2alpha_bwd(const float *__restrict__ exp_d, const float *__restrict__ grad_back,
                                float interval, const float *__restrict__ interval_arr, int64_t n,
                                float *__restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float itv = interval_arr ? interval_arr[i] : interval;
  const float ef = exp_d[i];
  const double e = (double)ef;
  const double em = e < 1e10 ? e : 1e10;
  const float pw = powf(1 + ef, -itv - 1);
  grad[i] = (float)(em * (double)pw * (double)itv * (double)grad_back[i]);
}

// ----------------------------------------------------------------------------------------------
// alpha -> weights: one 64-lane wave per ray.  Samples are loaded 64 at a time (coalesced); the
// transmittance recurrence T <- float(double(T) * (1 - double(alpha))) is evaluated in sample order
// on a wave-uniform value (every lane runs the same chain, lane k keeps step k's T), so rounding is
// identical to the reference's serial scan while loads/stores stay coalesced.  The wave stops the
// chain as soon as T < 1e-3 and only streams default values (w=0, T=1) over the rest of the ray.
// ----------------------------------------------------------------------------------------------
__global__ void k_segments(const int64_t *__restrict__ ray_id, int64_t n, int64_t *__restrict__ i_start,
                           int64_t *__restrict__ i_end) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t r = ray_id[i];
  if (i > 0) {
    const int64_t rp = ray_id[i - 1];
    if (r != rp) {
      i_start[r] = i;
      i_end[rp] = i;
    }
  }
  if (i == n - 1) i_end[r] = n;
}

__global__ void __launch_bounds__(256)
k_alpha2weight(const float *__restrict__ alpha, int64_t n_rays, float *__restrict__ weight,
               float *__restrict__ T, float *__restrict__ alphainv_last,
               const int64_t *__restrict__ i_start, int64_t *__restrict__ i_end) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= n_rays) return;
  const int lane = ug_lane();
  const int64_t i_s = i_start[r], i_e = i_end[r];
  float T_cum = 1.f;
  bool stopped = false;
  int64_t stop_at = i_e;
  for (int64_t base = i_s; base < i_e; base += UG_WAVE) {
    const int64_t i = base + lane;
    const int cnt = (int)((i_e - base) < UG_WAVE ? (i_e - base) : UG_WAVE);
    float myT = 1.f, myW = 0.f;
    if (!stopped) {
      const float a = (i < i_e) ? alpha[i] : 0.f;
      const double om = 1. - (double)a;
      for (int k = 0; k < cnt; ++k) {
        const float ak = ug_readlane_f(a, k);
        const double omk = ug_readlane_d(om, k);
        if (lane == k) {
          myT = T_cum;
          myW = T_cum * ak;
        }
        T_cum = (float)((double)T_cum * omk);
        if ((double)T_cum < 1e-3) {
          stopped = true;
          stop_at = base + k + 1;
          break;
        }
      }
      if (stopped && i >= stop_at) {
        myT = 1.f;
        myW = 0.f;
      }
    }
    if (i < i_e) {
      T[i] = myT;
      weight[i] = myW;
    }
  }
  if (lane == 0) {
    i_end[r] = stop_at;
    alphainv_last[r] = T_cum;
  }
}

// reverse pass: back_cum is a float running sum taken from the LAST kept sample backwards, so the
// chain again runs in order on a wave-uniform value; the per-sample double expression is lane-parallel.
__global__ void __launch_bounds__(256)
k_alpha2weight_bwd(const float *__restrict__ alpha, const float *__restrict__ weight,
                   const float *__restrict__ T, const float *__restrict__ alphainv_last,
                   const int64_t *__restrict__ i_start, const int64_t *__restrict__ i_end,
                   int64_t n_rays, const float *__restrict__ grad_weights,
                   const float *__restrict__ grad_last, float *__restrict__ grad) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= n_rays) return;
  const int lane = ug_lane();
  const int64_t i_s = i_start[r], i_e = i_end[r];
  float back = grad_last[r] * alphainv_last[r];
  for (int64_t top = i_e; top > i_s; top -= UG_WAVE) {
    // lane k holds sample top-1-k (reverse order inside the chunk)
    const int64_t i = top - 1 - lane;
    const bool ok = i >= i_s;
    const float gw = ok ? grad_weights[i] : 0.f;
    const float prod = ok ? gw * weight[i] : 0.f;
    const int cnt = (int)((top - i_s) < UG_WAVE ? (top - i_s) : UG_WAVE);
    float my_back = 0.f;
    for (int k = 0; k < cnt; ++k) {
      if (lane == k) my_back = back;
      back += ug_readlane_f(prod, k);
    }
    if (ok) {
      const float a = alpha[i];
      grad[i] = (float)((double)(gw * T[i]) - (double)my_back / ((double)(1 - a) + 1e-10));
    }
  }
}

// ----------------------------------------------------------------------------------------------
// total variation gradient (in place), dense or masked.  Quirk kept: x-axis term weighted by wz.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float ug_clamp1(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

template <bool DENSE>
__global__ void k_tv(const float *__restrict__ param, float *__restrict__ grad, float wy, float wz,
                     int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t N) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N) return;
  const float g0 = grad[idx];
  if (!(DENSE || g0 != 0.f)) return;
  const int64_t k = idx % sz_k;
  const int64_t j = idx / sz_k % sz_j;
  const int64_t i = idx / sz_k / sz_j % sz_i;
  const int64_t sj = sz_k, si = sz_k * sz_j;
  const float p = param[idx];
  // unconditional neighbour loads (a missing neighbour re-reads the element itself) with the missing terms switched off by a zero weight:
  // loads under their own exec branches are waited for one by one (see ug_tv_cl_one below); bit-identical (0 * clamp(p - p) = 0)
  const float n0 = param[idx - (k == 0 ? 0 : 1)], n1 = param[idx + (k == sz_k - 1 ? 0 : 1)];
  const float n2 = param[idx - (j == 0 ? 0 : sj)], n3 = param[idx + (j == sz_j - 1 ? 0 : sj)];
  const float n4 = param[idx - (i == 0 ? 0 : si)], n5 = param[idx + (i == sz_i - 1 ? 0 : si)];
  float g = 0;
  g += (k == 0 ? 0.f : wz) * ug_clamp1(p - n0);
  g += (k == sz_k - 1 ? 0.f : wz) * ug_clamp1(p - n1);
  g += (j == 0 ? 0.f : wy) * ug_clamp1(p - n2);
  g += (j == sz_j - 1 ? 0.f : wy) * ug_clamp1(p - n3);
  g += (i == 0 ? 0.f : wz) * ug_clamp1(p - n4);
  g += (i == sz_i - 1 ? 0.f : wz) * ug_clamp1(p - n5);
  grad[idx] = g0 + g;
}

// XCD = blocks renumbered so that each of the 8 XCDs (the hardware deals consecutive workgroups to them round-robin)
// walks one contiguous eighth of the array: the stencil's j / i neighbour lines are then found in the XCD's own L2
// instead of being fetched again over the fabric by another one (4.98 -> 4.41 ms on the channel-last S3 k0 array).
typedef float ug_v4f __attribute__((ext_vector_type(4)));
// streaming arrays (gradient, moments, the new parameters) bypass the caches' retention: the L2 is left to the stencil
template <bool NT> __device__ __forceinline__ float4 ug_ld4(const float *p) {
  if (NT) { const ug_v4f v = __builtin_nontemporal_load((const ug_v4f *)p); return make_float4(v.x, v.y, v.z, v.w); }
  return *(const float4 *)p;
}
template <bool NT> __device__ __forceinline__ void ug_st4(float *p, float a, float b, float c, float d) {
  if (NT) { ug_v4f v = {a, b, c, d}; __builtin_nontemporal_store(v, (ug_v4f *)p); }
  else *(float4 *)p = make_float4(a, b, c, d);
}

// true for all 8 lanes of a 128-byte line (8 consecutive float4 lanes) when any of them says so: whole-line stores
__device__ __forceinline__ bool ug_line_any(bool mine) {
  const unsigned long long m = __ballot(mine);
  return ((m >> (__lane_id() & ~7u)) & 0xFFull) != 0;
}

// touched-line bitmap of a recycled gradient buffer (k_grid_query_backward): float4 lane q belongs to the 256-byte line
// q >> 4; null = no bitmap, every line counts as touched
__device__ __forceinline__ bool ug_touched(const uint32_t *__restrict__ touch, unsigned q) {
  return !touch || ((touch[q >> 9] >> ((q >> 4) & 31u)) & 1u);
}

template <int XCD>
__device__ __forceinline__ unsigned ug_xcd_block() {
  unsigned b = blockIdx.x;
  if (XCD) {
    const unsigned nb = gridDim.x, per = nb >> 3, rem = nb & 7u, xcd = b & 7u;
    b = xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
  }
  return b;
}

// 4 voxels (one float4 along the fastest axis) per lane, 32-bit index arithmetic: used when sz_k % 4 == 0,
// N < 2^31 and both arrays are 16-byte aligned.  The per-voxel expression (six sequential float adds) is the
// scalar kernel's, so results are bit-identical; neighbours along k come from the same float4 plus two scalar
// loads, neighbours along j / i are four more float4 loads.
template <bool DENSE, int XCD = 0>
__global__ void __launch_bounds__(256)
k_tv_vec4(const float *__restrict__ param, float *__restrict__ grad, float wy, float wz, int sz_i, int sz_j,
          int sz_k, unsigned n4) {
  const unsigned q = ug_xcd_block<XCD>() * blockDim.x + threadIdx.x;
  if (q >= n4) return;
  const unsigned idx = q * 4u;
  float4 g0 = *(const float4 *)(grad + idx);
  if (!DENSE && g0.x == 0.f && g0.y == 0.f && g0.z == 0.f && g0.w == 0.f) return;
  const unsigned k4 = (unsigned)sz_k >> 2;
  const unsigned kq = q % k4, row = q / k4;      // row = (plane * sz_i + i) * sz_j + j
  const unsigned j = row % (unsigned)sz_j, i = (row / (unsigned)sz_j) % (unsigned)sz_i;
  const unsigned sj = (unsigned)sz_k, si = (unsigned)sz_k * (unsigned)sz_j;
  const float4 p = *(const float4 *)(param + idx);
  const bool k_first = kq == 0, k_last = kq == k4 - 1;
  // unconditional neighbour loads (a missing neighbour re-reads a value of the element itself), the missing terms switched off by a
  // zero weight: loads under their own exec branches are waited for one by one (see ug_tv_cl_one); bit-identical
  const float pm = param[idx - (k_first ? 0u : 1u)];
  const float pp = param[idx + (k_last ? 3u : 4u)];
  const float wkm = k_first ? 0.f : wz, wkp = k_last ? 0.f : wz;
  const float wj0 = j != 0 ? wy : 0.f, wj1 = j != (unsigned)sz_j - 1 ? wy : 0.f;
  const float wi0 = i != 0 ? wz : 0.f, wi1 = i != (unsigned)sz_i - 1 ? wz : 0.f;
  const float4 nj0 = *(const float4 *)(param + idx - (j != 0 ? sj : 0u));
  const float4 nj1 = *(const float4 *)(param + idx + (j != (unsigned)sz_j - 1 ? sj : 0u));
  const float4 ni0 = *(const float4 *)(param + idx - (i != 0 ? si : 0u));
  const float4 ni1 = *(const float4 *)(param + idx + (i != (unsigned)sz_i - 1 ? si : 0u));
  const float pv[4] = {p.x, p.y, p.z, p.w}, gv[4] = {g0.x, g0.y, g0.z, g0.w};
  const float km[4] = {pm, p.x, p.y, p.z}, kp[4] = {p.y, p.z, p.w, pp};
  const float a0[4] = {nj0.x, nj0.y, nj0.z, nj0.w}, a1[4] = {nj1.x, nj1.y, nj1.z, nj1.w};
  const float b0[4] = {ni0.x, ni0.y, ni0.z, ni0.w}, b1[4] = {ni1.x, ni1.y, ni1.z, ni1.w};
  float out[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float g = 0;
    g += (e == 0 ? wkm : wz) * ug_clamp1(pv[e] - km[e]);
    g += (e == 3 ? wkp : wz) * ug_clamp1(pv[e] - kp[e]);
    g += wj0 * ug_clamp1(pv[e] - a0[e]);
    g += wj1 * ug_clamp1(pv[e] - a1[e]);
    g += wi0 * ug_clamp1(pv[e] - b0[e]);
    g += wi1 * ug_clamp1(pv[e] - b1[e]);
    out[e] = (DENSE || gv[e] != 0.f) ? gv[e] + g : gv[e];
  }
  *(float4 *)(grad + idx) = make_float4(out[0], out[1], out[2], out[3]);
}

// ----------------------------------------------------------------------------------------------
// cumdist_thres: one wave per ray, 64 distances per coalesced load, serial float chain as above
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_cumdist(const float *__restrict__ dist, float thres, int64_t n_rays, int64_t n_pts,
          uint8_t *__restrict__ mask) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= n_rays) return;
  const int lane = ug_lane();
  float cum = 0.f;
  for (int64_t base = 0; base < n_pts; base += UG_WAVE) {
    const int64_t i = base + lane;
    const float d = (i < n_pts) ? dist[r * n_pts + i] : 0.f;
    const int cnt = (int)((n_pts - base) < UG_WAVE ? (n_pts - base) : UG_WAVE);
    bool my_over = false;
    for (int k = 0; k < cnt; ++k) {
      cum += ug_readlane_f(d, k);
      const bool over = cum > thres;
      if (lane == k) my_over = over;
      cum *= over ? 0.f : 1.f;
    }
    if (i < n_pts) mask[r * n_pts + i] = (uint8_t)my_over;
  }
}

// ----------------------------------------------------------------------------------------------
// segment_cumsum (called by the reference's DistortionLoss, FourierGrid_model.py:684-708, never shipped by its
// ub360_utils.cpp): exclusive running sums of w and w*s inside each ray segment + per-ray totals.  One wave per
// ray, 64 samples per coalesced load, the two fp32 chains run in sample order on wave-uniform values.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_segment_cumsum(const float *__restrict__ w, const float *__restrict__ s, int64_t n_rays,
                 const int64_t *__restrict__ i_start, const int64_t *__restrict__ i_end,
                 float *__restrict__ w_prefix, float *__restrict__ w_total, float *__restrict__ ws_prefix,
                 float *__restrict__ ws_total) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= n_rays) return;
  const int lane = ug_lane();
  const int64_t i_s = i_start[r], i_e = i_end[r];
  float cw = 0.f, cws = 0.f;
  for (int64_t base = i_s; base < i_e; base += UG_WAVE) {
    const int64_t i = base + lane;
    const float wi = (i < i_e) ? w[i] : 0.f;
    const float wsi = (i < i_e) ? wi * s[i] : 0.f;
    const int cnt = (int)((i_e - base) < UG_WAVE ? (i_e - base) : UG_WAVE);
    float my_w = 0.f, my_ws = 0.f;
    for (int k = 0; k < cnt; ++k) {
      if (lane == k) { my_w = cw; my_ws = cws; }
      cw = cw + ug_readlane_f(wi, k);
      cws = cws + ug_readlane_f(wsi, k);
    }
    if (i < i_e) { w_prefix[i] = my_w; ws_prefix[i] = my_ws; }
  }
  if (lane == 0) { w_total[r] = cw; ws_total[r] = cws; }
}

// ----------------------------------------------------------------------------------------------
// Adam family.  MODE 0 dense, 1 masked (skip grad==0), 2 per-voxel lr.  4 voxels per lane with
// 16-byte loads; in masked mode a lane touches m/v/param only when one of its 4 grads is non-zero,
// so an almost-empty gradient costs ~4 B/voxel of HBM reads.
// ----------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void ug_adam_one(float &p, float g, float &m, float &v, float lrk,
                                            float step_size, float beta1, float beta2, float eps) {
  m = beta1 * m + (1 - beta1) * g;
  v = beta2 * v + (1 - beta2) * g * g;
  if (MODE == 2) p -= step_size * lrk * m / (sqrtf(v) + eps);
  else p -= step_size * m / (sqrtf(v) + eps);
}

// RZ (masked mode only): the gradient is overwritten with zeros after use, whole 128-byte lines at a time and only
// those that held something -- the buffer goes back to the zero pool of the grid's backward (_gradpool.py)
template <int MODE, bool RZ>
__device__ __forceinline__ void ug_adam_vec4_one(float4 *__restrict__ param, const float4 *__restrict__ grad, float4 *__restrict__ exp_avg,
                                                 float4 *__restrict__ exp_avg_sq, const float4 *__restrict__ perlr, int64_t i,
                                                 float step_size, float beta1, float beta2, float eps) {
  const float4 g = grad[i];
  if (RZ && ug_line_any(g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f))
    const_cast<float4 *>(grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 1 && g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) return;
  float4 p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
  float4 l = make_float4(1.f, 1.f, 1.f, 1.f);
  if (MODE == 2) l = perlr[i];
  if (MODE != 1 || g.x != 0.f) ug_adam_one<MODE>(p.x, g.x, m.x, v.x, l.x, step_size, beta1, beta2, eps);
  if (MODE != 1 || g.y != 0.f) ug_adam_one<MODE>(p.y, g.y, m.y, v.y, l.y, step_size, beta1, beta2, eps);
  if (MODE != 1 || g.z != 0.f) ug_adam_one<MODE>(p.z, g.z, m.z, v.z, l.z, step_size, beta1, beta2, eps);
  if (MODE != 1 || g.w != 0.f) ug_adam_one<MODE>(p.w, g.w, m.w, v.w, l.w, step_size, beta1, beta2, eps);
  param[i] = p;
  exp_avg[i] = m;
  exp_avg_sq[i] = v;
}

template <int MODE, bool RZ = false>
__global__ void __launch_bounds__(256)
k_adam_vec4(float4 *__restrict__ param, const float4 *__restrict__ grad, float4 *__restrict__ exp_avg,
            float4 *__restrict__ exp_avg_sq, const float4 *__restrict__ perlr, int64_t n4,
            float step_size, float beta1, float beta2, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x)
    ug_adam_vec4_one<MODE, RZ>(param, grad, exp_avg, exp_avg_sq, perlr, i, step_size, beta1, beta2, eps);
}

// Walk over the touched-line bitmap of a recycled gradient (k_grid_query_backward).  One wave owns 64 consecutive 32-bit words
// (= 2 048 lines of 256 bytes): every lane fetches one word, a ballot finds the words that hold anything, and the wave visits those in
// turn -- a word's set bits dealt to the wave's four 16-lane quarters, so a word with <= 4 marked lines costs one round.  `q` handed to
// the body is the float4 index of the lane.  (Until round 6 a wave owned ONE word: at a few per cent of the lines marked most of the
// 4e5 waves of S3's k0 grid lived for one load and an exit, and the kernel's time was its wave count times a memory latency.)
// wpw = words per wave (1..64): ug_touch_wpw keeps >= ~16 k waves in the launch (a small grid must not be walked by a handful of waves)
static inline int ug_touch_wpw(int64_t n_words) {
  int w = 64;
  while (w > 1 && n_words / w < 16384) w >>= 1;
  return w;
}
#define UG_TOUCH_WALK(touch, n_words, n4, wpw, BODY)                                                                   \
  {                                                                                                                    \
    const int64_t w0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * (wpw);                                \
    if (w0 >= (n_words)) return;                                                                                       \
    const int64_t wi = w0 + ug_lane();                                                                                 \
    const bool have = ug_lane() < (wpw) && wi < (n_words);                                                             \
    const uint32_t mine = (touch)[have ? wi : w0] * (have ? 1u : 0u);                                                  \
    unsigned long long nz = __ballot(mine != 0u);                                                                      \
    const int quarter = ug_lane() >> 4;                                                                                \
    while (nz != 0ull) {                                                                                               \
      const int src = __builtin_ctzll(nz);                                                                             \
      nz &= nz - 1ull;                                                                                                 \
      const uint32_t mask = (uint32_t)__builtin_amdgcn_readlane((int)mine, src);                                       \
      const int64_t word = w0 + src;                                                                                   \
      const int cnt = __popc(mask);                                                                                    \
      for (int base = 0; base < cnt; base += 4) {                                                                      \
        const int k = base + quarter;                                                                                  \
        uint32_t m = mask;                                                                                             \
        for (int i = 0; i < k; ++i) m &= m - 1u;                                                                       \
        const int64_t q = ((word << 5) + (int64_t)(__ffs(m) - 1)) * 16 + (ug_lane() & 15);                             \
        if (k < cnt && q < (int64_t)(n4)) { BODY }                                                                     \
      }                                                                                                                \
    }                                                                                                                  \
  }

// masked Adam on the marked lines only; the gradient comes back all zero (RZ)
__global__ void __launch_bounds__(256)
k_adam_vec4_touch(float4 *__restrict__ param, const float4 *__restrict__ grad, float4 *__restrict__ exp_avg,
                  float4 *__restrict__ exp_avg_sq, int64_t n4, float step_size, float beta1, float beta2, float eps,
                  const uint32_t *__restrict__ touch, int64_t n_words, int wpw) {
  UG_TOUCH_WALK(touch, n_words, n4, wpw, (ug_adam_vec4_one<1, true>(param, grad, exp_avg, exp_avg_sq, nullptr, q, step_size, beta1, beta2, eps));)
}

template <int MODE, bool RZ = false>
__global__ void k_adam_scalar(float *__restrict__ param, const float *__restrict__ grad,
                              float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                              const float *__restrict__ perlr, int64_t begin, int64_t N,
                              float step_size, float beta1, float beta2, float eps) {
  const int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float g = grad[i];
  if (RZ && g != 0.f) const_cast<float *>(grad)[i] = 0.f;
  if (MODE == 1 && !(g != 0.f)) return;
  float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
  ug_adam_one<MODE>(p, g, m, v, MODE == 2 ? perlr[i] : 1.f, step_size, beta1, beta2, eps);
  param[i] = p;
  exp_avg[i] = m;
  exp_avg_sq[i] = v;
}

template <int MODE, bool RZ = false>
static int ug_adam_launch(float *param, const float *grad, float *m, float *v, const float *perlr,
                          int64_t N, float step_size, float b1, float b2, float eps, hipStream_t st) {
  const uintptr_t al = (uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v |
                       (MODE == 2 ? (uintptr_t)perlr : 0);
  int64_t done = 0;
  if ((al & 15) == 0 && N >= 4) {
    const int64_t n4 = N / 4;
    // one float4 per lane, no grid-stride loop: measured against the reference's own one-element-per-thread kernels on
    // the same MI355X (tools/bench_dropin_ops.py), a capped grid of 32 blocks per CU streamed 672 M voxels at
    // 5.5 TB/s where the plain huge grid reaches > 6 TB/s (the loop only serialises independent 16-byte streams)
    const int64_t blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_adam_vec4<MODE, RZ>), dim3((unsigned)blocks), dim3(256), 0, st,
                       (float4 *)param, (const float4 *)grad, (float4 *)m, (float4 *)v,
                       (const float4 *)perlr, n4, step_size, b1, b2, eps);
    done = n4 * 4;
  }
  if (done < N)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_adam_scalar<MODE, RZ>), dim3(ug_blocks(N - done, 256)), dim3(256), 0,
                       st, param, grad, m, v, perlr, done, N, step_size, b1, b2, eps);
  UG_LAUNCH_CHECK();
  return 0;
}


// ----------------------------------------------------------------------------------------------
// Fused DENSE total-variation gradient + Adam (new entry point, no reference counterpart).
// While `tv_dense_before` holds (run_train.py:281-287, 10 000 of truck_single's 30 000 iterations) the reference runs
// total_variation_add_grad(dense) -- which makes EVERY gradient entry non-zero -- and then masked_adam_upd, i.e. two
// full passes over param / grad and one over both moments: 13 arrays of traffic.  Fused: per 4 voxels the TV term is
// formed exactly as k_tv_vec4<true> does (same six sequential adds), added to the gradient in registers and fed to
// ug_adam_one: 7 arrays (param, grad, m, v read; param', m, v written), the gradient is never written back.  The
// stencil needs the neighbours' OLD values, so the new parameters go to a second buffer that the caller swaps in.
// Bit-identical to the two-kernel sequence.  MASKED = the skip_zero_grad rule applied to the TV-added gradient.
// ----------------------------------------------------------------------------------------------
template <bool MASKED, int XCD = 0>
__global__ void __launch_bounds__(256)
k_tv_adam_vec4(const float *__restrict__ param, float *__restrict__ param_out, const float *__restrict__ grad,
               float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, float wy, float wz, int sz_i, int sz_j,
               int sz_k, unsigned n4, float step_size, float beta1, float beta2, float eps, int rezero) {
  const unsigned q = ug_xcd_block<XCD>() * blockDim.x + threadIdx.x;
  if (q >= n4) return;
  const unsigned idx = q * 4u;
  const float4 g0 = ug_ld4<XCD == 2>(grad + idx);
  // rezero: the gradient buffer goes back to the zero pool (_gradpool.py) -- only the touched 128-byte lines are written
  if (rezero && ug_line_any(g0.x != 0.f || g0.y != 0.f || g0.z != 0.f || g0.w != 0.f))
    *(float4 *)(const_cast<float *>(grad) + idx) = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned k4 = (unsigned)sz_k >> 2;
  const unsigned kq = q % k4, row = q / k4;
  const unsigned j = row % (unsigned)sz_j, i = (row / (unsigned)sz_j) % (unsigned)sz_i;
  const unsigned sj = (unsigned)sz_k, si = (unsigned)sz_k * (unsigned)sz_j;
  const float4 p = *(const float4 *)(param + idx);
  const bool k_first = kq == 0, k_last = kq == k4 - 1;
  // unconditional neighbour loads (a missing neighbour re-reads a value of the element itself), the missing terms switched off by a
  // zero weight: loads under their own exec branches are waited for one by one (see ug_tv_cl_one); bit-identical
  const float pm = param[idx - (k_first ? 0u : 1u)];
  const float pp = param[idx + (k_last ? 3u : 4u)];
  const float wkm = k_first ? 0.f : wz, wkp = k_last ? 0.f : wz;
  const float wj0 = j != 0 ? wy : 0.f, wj1 = j != (unsigned)sz_j - 1 ? wy : 0.f;
  const float wi0 = i != 0 ? wz : 0.f, wi1 = i != (unsigned)sz_i - 1 ? wz : 0.f;
  const float4 nj0 = *(const float4 *)(param + idx - (j != 0 ? sj : 0u));
  const float4 nj1 = *(const float4 *)(param + idx + (j != (unsigned)sz_j - 1 ? sj : 0u));
  const float4 ni0 = *(const float4 *)(param + idx - (i != 0 ? si : 0u));
  const float4 ni1 = *(const float4 *)(param + idx + (i != (unsigned)sz_i - 1 ? si : 0u));
  const float4 m4 = ug_ld4<XCD == 2>(exp_avg + idx), v4 = ug_ld4<XCD == 2>(exp_avg_sq + idx);
  float pv[4] = {p.x, p.y, p.z, p.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
  const float pold[4] = {p.x, p.y, p.z, p.w}, gv[4] = {g0.x, g0.y, g0.z, g0.w};
  const float km[4] = {pm, p.x, p.y, p.z}, kp[4] = {p.y, p.z, p.w, pp};
  const float a0[4] = {nj0.x, nj0.y, nj0.z, nj0.w}, a1[4] = {nj1.x, nj1.y, nj1.z, nj1.w};
  const float b0[4] = {ni0.x, ni0.y, ni0.z, ni0.w}, b1[4] = {ni1.x, ni1.y, ni1.z, ni1.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float g = 0;
    g += (e == 0 ? wkm : wz) * ug_clamp1(pold[e] - km[e]);
    g += (e == 3 ? wkp : wz) * ug_clamp1(pold[e] - kp[e]);
    g += wj0 * ug_clamp1(pold[e] - a0[e]);
    g += wj1 * ug_clamp1(pold[e] - a1[e]);
    g += wi0 * ug_clamp1(pold[e] - b0[e]);
    g += wi1 * ug_clamp1(pold[e] - b1[e]);
    const float gt = gv[e] + g;
    if (!MASKED || gt != 0.f) ug_adam_one<0>(pv[e], gt, mv[e], vv[e], 1.f, step_size, beta1, beta2, eps);
  }
  ug_st4<XCD == 2>(param_out + idx, pv[0], pv[1], pv[2], pv[3]);
  ug_st4<XCD == 2>(exp_avg + idx, mv[0], mv[1], mv[2], mv[3]);
  ug_st4<XCD == 2>(exp_avg_sq + idx, vv[0], vv[1], vv[2], vv[3]);
}


// ----------------------------------------------------------------------------------------------
// get_rays_of_a_view (dvgo.py:493-521,554-559) as ONE kernel: pixel-centre pinhole rays,
// rays_d = dirs . c2w[:3,:3]^T, viewdirs = rays_d / |rays_d|, rays_o = c2w[:,3]; the torch chain it replaces is ~15
// launches per frame.  Operation order of the reference's elementwise chain (products, then ((p0 + p1) + p2); the norm as
// the fma chain of torch's vector_norm).  pix == nullptr: all H*W pixels in image order; else the listed flat indices.
// ----------------------------------------------------------------------------------------------
struct ug_cam { float fx, fy, cx, cy; int W, H, inverse_y, flip_x, flip_y, center; };

__global__ void k_rays_of_a_view(ug_cam c, const float *__restrict__ c2w, const int64_t *__restrict__ pix, int64_t n,
                                 float *__restrict__ rays_o, float *__restrict__ rays_d, float *__restrict__ viewdirs) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int64_t p = pix ? pix[t] : t;
  int j = (int)(p / c.W), i = (int)(p - (int64_t)j * c.W);
  if (c.flip_x) i = c.W - 1 - i;
  if (c.flip_y) j = c.H - 1 - j;
  float ii = (float)i, jj = (float)j;
  if (c.center) { ii = ii + 0.5f; jj = jj + 0.5f; }
  float dx, dy, dz;
  if (c.inverse_y) { dx = (ii - c.cx) / c.fx; dy = (jj - c.cy) / c.fy; dz = 1.0f; }
  else { dx = (ii - c.cx) / c.fx; dy = -(jj - c.cy) / c.fy; dz = -1.0f; }
  float r[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p0 = dx * c2w[4 * a], p1 = dy * c2w[4 * a + 1], p2 = dz * c2w[4 * a + 2];
    r[a] = (p0 + p1) + p2;
  }
  const float nrm = sqrtf(fmaf(r[2], r[2], fmaf(r[1], r[1], r[0] * r[0])));
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    rays_o[3 * t + a] = c2w[4 * a + 3];
    rays_d[3 * t + a] = r[a];
    viewdirs[3 * t + a] = r[a] / nrm;
  }
}


// ----------------------------------------------------------------------------------------------
// Channel-last ([P][X][Y][Z][C], torch channels_last_3d) variants of the TV gradient and of the fused dense TV + Adam
// pass: one lane = 4 channels of a voxel (C % 4 == 0), neighbours of the SAME channels at +-C (k), +-Z*C (j), +-Y*Z*C
// (i).  Per element the expression is the canonical kernels' (six sequential adds, the wz-for-x quirk), so the results
// equal the canonical-layout results element for element.  ADAM: 0 = TV only (grad updated in place), 1 = fused with
// masked Adam, 2 = fused with dense Adam (param_out written, grad untouched).
// ----------------------------------------------------------------------------------------------
template <bool DENSE, int ADAM, int XCD>
__device__ __forceinline__ void ug_tv_cl_one(const float *__restrict__ param, float *__restrict__ param_out, float *__restrict__ grad,
                                             float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, float wy, float wz, int sz_i,
                                             int sz_j, int sz_k, int C, unsigned q, float step_size, float beta1, float beta2, float eps,
                                             int rezero, bool hit) {
  const unsigned idx = q * 4u;
  // hit = false: an unmarked line of a recycled gradient is all zero and is not read (dense mode only; the masked mode
  // does not come here for such a line)
  const float4 g0 = hit ? ug_ld4<XCD == 2>(grad + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (!DENSE && g0.x == 0.f && g0.y == 0.f && g0.z == 0.f && g0.w == 0.f) return;
  const unsigned c4 = (unsigned)C >> 2;
  const unsigned vox = q / c4;                                  // (plane * sz_i + i) * sz_j * sz_k + j * sz_k + k
  const unsigned k = vox % (unsigned)sz_k, j = (vox / (unsigned)sz_k) % (unsigned)sz_j,
                 i = (vox / ((unsigned)sz_k * (unsigned)sz_j)) % (unsigned)sz_i;
  const unsigned sk = (unsigned)C, sj = (unsigned)sz_k * sk, si = (unsigned)sz_j * sj;
  const float4 p = *(const float4 *)(param + idx);
  // the six neighbours by UNCONDITIONAL loads (a missing neighbour re-reads the element itself) and their terms switched off by a zero
  // WEIGHT below: with `if (k != 0) nk0 = load` every load sat under its own exec branch and hipcc waited vmcnt(0) behind each -- seven
  // round trips per element one after the other (round 6).  Bit-identical: a switched-off term is 0 * clamp(p - p) = 0, as before.
  const float wk0 = k != 0 ? wz : 0.f, wk1 = k != (unsigned)sz_k - 1 ? wz : 0.f;
  const float wj0 = j != 0 ? wy : 0.f, wj1 = j != (unsigned)sz_j - 1 ? wy : 0.f;
  const float wi0 = i != 0 ? wz : 0.f, wi1 = i != (unsigned)sz_i - 1 ? wz : 0.f;
  const float4 nk0 = *(const float4 *)(param + idx - (k != 0 ? sk : 0u));
  const float4 nk1 = *(const float4 *)(param + idx + (k != (unsigned)sz_k - 1 ? sk : 0u));
  const float4 nj0 = *(const float4 *)(param + idx - (j != 0 ? sj : 0u));
  const float4 nj1 = *(const float4 *)(param + idx + (j != (unsigned)sz_j - 1 ? sj : 0u));
  const float4 ni0 = *(const float4 *)(param + idx - (i != 0 ? si : 0u));
  const float4 ni1 = *(const float4 *)(param + idx + (i != (unsigned)sz_i - 1 ? si : 0u));
  float pv[4] = {p.x, p.y, p.z, p.w};
  const float pold[4] = {p.x, p.y, p.z, p.w}, gv[4] = {g0.x, g0.y, g0.z, g0.w};
  const float k0[4] = {nk0.x, nk0.y, nk0.z, nk0.w}, k1[4] = {nk1.x, nk1.y, nk1.z, nk1.w};
  const float a0[4] = {nj0.x, nj0.y, nj0.z, nj0.w}, a1[4] = {nj1.x, nj1.y, nj1.z, nj1.w};
  const float b0[4] = {ni0.x, ni0.y, ni0.z, ni0.w}, b1[4] = {ni1.x, ni1.y, ni1.z, ni1.w};
  float mv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};
  if (ADAM) {
    const float4 m4 = ug_ld4<XCD == 2>(exp_avg + idx), v4 = ug_ld4<XCD == 2>(exp_avg_sq + idx);
    mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
    vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
  }
  float out[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float g = 0;
    g += wk0 * ug_clamp1(pold[e] - k0[e]);
    g += wk1 * ug_clamp1(pold[e] - k1[e]);
    g += wj0 * ug_clamp1(pold[e] - a0[e]);
    g += wj1 * ug_clamp1(pold[e] - a1[e]);
    g += wi0 * ug_clamp1(pold[e] - b0[e]);
    g += wi1 * ug_clamp1(pold[e] - b1[e]);
    out[e] = (DENSE || gv[e] != 0.f) ? gv[e] + g : gv[e];
    if (ADAM) {
      if (ADAM == 2 || out[e] != 0.f) ug_adam_one<0>(pv[e], out[e], mv[e], vv[e], 1.f, step_size, beta1, beta2, eps);
    }
  }
  if (ADAM) {
    ug_st4<XCD == 2>(param_out + idx, pv[0], pv[1], pv[2], pv[3]);
    ug_st4<XCD == 2>(exp_avg + idx, mv[0], mv[1], mv[2], mv[3]);
    ug_st4<XCD == 2>(exp_avg_sq + idx, vv[0], vv[1], vv[2], vv[3]);
    if (rezero && ug_line_any(g0.x != 0.f || g0.y != 0.f || g0.z != 0.f || g0.w != 0.f))
      *(float4 *)(grad + idx) = make_float4(0.f, 0.f, 0.f, 0.f);     // (a 128-byte line lies inside one 256-byte bitmap line)
  } else {
    *(float4 *)(grad + idx) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

template <bool DENSE, int ADAM, int XCD = 0>
__global__ void __launch_bounds__(256)
k_tv_cl_vec4(const float *__restrict__ param, float *__restrict__ param_out, float *__restrict__ grad,
             float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, float wy, float wz, int sz_i, int sz_j,
             int sz_k, int C, unsigned n4, float step_size, float beta1, float beta2, float eps, int rezero,
             const uint32_t *__restrict__ touch) {
  const unsigned q = ug_xcd_block<XCD>() * blockDim.x + threadIdx.x;
  if (q >= n4) return;
  ug_tv_cl_one<DENSE, ADAM, XCD>(param, param_out, grad, exp_avg, exp_avg_sq, wy, wz, sz_i, sz_j, sz_k, C, q, step_size, beta1, beta2,
                                 eps, rezero, ug_touched(touch, q));
}

// SLAB ORDER of the fused dense pass (round 5, tv_xcd = 3).  The linear walk fetches every i-plane of the parameter about THREE times
// from memory: the i-1 / i+1 neighbours of a voxel are a whole plane away (Y x Z x C x 4 B = 1.9 MB at S3's k0 grid), a parameter
// line would have to survive two plane-times in an XCD's 4 MB L2 beside four streaming arrays, and the request counters show it
// does not -- 20.7 GB read per launch where 13.8 GB are needed, 31.0 GB moved in 4.31 ms = 7.2 TB/s of fabric traffic for 5.6 TB/s of
// useful bytes (profiles/r05/tv_adam_dense_pmc.txt).  Here the SAME one-float4-per-lane kernel visits the array in slabs of JW rows
// of j: workgroup b -> (level, slab, i, chunk of the slab's row run) with the chunk fastest, then i, then the slab -- three
// consecutive i-planes of a slab are 3 x JW x Z x C x 4 B = 720 KB and stay in L2, a slab's two boundary rows are the only lines
// read twice (8 %).  Same loads, same expression per element: bit-identical results.  (A variant that walked along i inside a
// workgroup with the three centre values in registers cut the reads to 14.2 GB as well but ran 10 % SLOWER: every step of every
// resident workgroup jumped 1.9 MB in seven arrays -- profiles/r05/tv_adam_dense_ab.txt.)
struct ug_tv_slab { unsigned jw, n_slab, blocks_per_row_run, row4; };      // row4 = Z x C / 4 float4 per j-row

template <int ADAM>
__global__ void __launch_bounds__(256)
k_tv_cl_slab(const float *__restrict__ param, float *__restrict__ param_out, float *__restrict__ grad,
             float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, float wy, float wz, int sz_i, int sz_j,
             int sz_k, int C, unsigned n4, ug_tv_slab sl, float step_size, float beta1, float beta2, float eps, int rezero,
             const uint32_t *__restrict__ touch) {
  const unsigned b = ug_xcd_block<1>();
  const unsigned chunk = b % sl.blocks_per_row_run, r1 = b / sl.blocks_per_row_run;
  const unsigned i = r1 % (unsigned)sz_i, r2 = r1 / (unsigned)sz_i;
  const unsigned slab = r2 % sl.n_slab, level = r2 / sl.n_slab;
  const unsigned j0 = slab * sl.jw, rows = min(sl.jw, (unsigned)sz_j - j0);
  const unsigned within = chunk * 256u + threadIdx.x;
  if (within >= rows * sl.row4) return;
  const unsigned q = ((level * (unsigned)sz_i + i) * (unsigned)sz_j + j0) * sl.row4 + within;
  if (q >= n4) return;
  ug_tv_cl_one<true, ADAM, 2>(param, param_out, grad, exp_avg, exp_avg_sq, wy, wz, sz_i, sz_j, sz_k, C, q, step_size, beta1, beta2,
                              eps, rezero, ug_touched(touch, q));
}

// masked TV gradient on the marked lines of a recycled gradient (UG_TOUCH_WALK)
__global__ void __launch_bounds__(256)
k_tv_cl_touch(const float *__restrict__ param, float *__restrict__ grad, float wy, float wz, int sz_i, int sz_j, int sz_k, int C,
              unsigned n4, const uint32_t *__restrict__ touch, int64_t n_words, int wpw) {
  UG_TOUCH_WALK(touch, n_words, n4, wpw, (ug_tv_cl_one<false, 0, 0>(param, nullptr, grad, nullptr, nullptr, wy, wz, sz_i, sz_j, sz_k, C,
                                                               (unsigned)q, 0.f, 0.f, 0.f, 0.f, 0, true));)
}

static int g_tv_xcd = 3;   // ugrid_tune("tv_xcd", 0|1|2|3): dense TV (+ Adam) kernels: linear block order | XCD-contiguous | + non-temporal
                           // streams | + slab order of the fused channel-last pass (k_tv_cl_slab, default)
extern "C" int ug_set_tv_xcd(int m) { if (m < 0 || m > 3) return 1; g_tv_xcd = m; return 0; }

// ----------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------
#define ST(s) ((hipStream_t)(s))

extern "C" int ugrid_infer_t_minmax(const float *rays_o, const float *rays_d, const float *xyz_min,
                                    const float *xyz_max, float near, float far, int64_t n_rays,
                                    float *t_min, float *t_max, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(k_infer_t_minmax, dim3(ug_blocks(n_rays, 256)), dim3(256), 0, ST(s), rays_o, rays_d,
                     xyz_min, xyz_max, near, far, n_rays, t_min, t_max);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_infer_n_samples(const float *rays_d, const float *t_min, const float *t_max,
                                     float stepdist, int64_t n_rays, int64_t *n_samples, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(k_infer_n_samples, dim3(ug_blocks(n_rays, 256)), dim3(256), 0, ST(s), rays_d, t_min,
                     t_max, stepdist, n_rays, n_samples);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_infer_ray_start_dir(const float *rays_o, const float *rays_d, const float *t_min,
                                         int64_t n_rays, float *rays_start, float *rays_dir,
                                         ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(k_infer_ray_start_dir, dim3(ug_blocks(n_rays, 256)), dim3(256), 0, ST(s), rays_o,
                     rays_d, t_min, n_rays, rays_start, rays_dir);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_sample_pts_on_rays_count(const float *rays_o, const float *rays_d,
                                              const float *xyz_min, const float *xyz_max, float near,
                                              float far, float stepdist, int64_t n_rays, float *t_min,
                                              float *t_max, int64_t *n_steps, int64_t *n_steps_cumsum,
                                              int64_t *d_total, void *scan_ws, ugrid_stream_t s) {
  if (n_rays <= 0) return (int)hipMemsetAsync(d_total, 0, sizeof(int64_t), ST(s));
  hipLaunchKernelGGL(k_sample_count, dim3(ug_blocks(n_rays, 256)), dim3(256), 0, ST(s), rays_o, rays_d,
                     xyz_min, xyz_max, near, far, stepdist, n_rays, t_min, t_max, n_steps);
  UG_LAUNCH_CHECK();
  return ug_inclusive_scan(n_steps, n_rays, n_steps_cumsum, d_total, scan_ws, ST(s));
}

extern "C" int ugrid_sample_pts_on_rays_fill(const float *rays_o, const float *rays_d,
                                             const float *xyz_min, const float *xyz_max,
                                             const float *t_min, const int64_t *n_steps_cumsum,
                                             float stepdist, int64_t n_rays, int64_t total_len,
                                             float *rays_pts, uint8_t *mask_outbbox, int64_t *ray_id,
                                             int64_t *step_id, ugrid_stream_t s) {
  if (total_len <= 0 || n_rays <= 0) return 0;
  hipLaunchKernelGGL(k_sample_fill, dim3(ug_blocks(total_len, 256)), dim3(256), 0, ST(s), rays_o, rays_d,
                     xyz_min, xyz_max, t_min, n_steps_cumsum, stepdist, n_rays, total_len, rays_pts,
                     mask_outbbox, ray_id, step_id);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_sample_ndc_pts_on_rays(const float *rays_o, const float *rays_d,
                                            const float *xyz_min, const float *xyz_max,
                                            int64_t n_samples, int64_t n_rays, float *rays_pts,
                                            uint8_t *mask_outbbox, ugrid_stream_t s) {
  const int64_t total = n_samples * n_rays;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(k_sample_ndc, dim3(ug_blocks(total, 256)), dim3(256), 0, ST(s), rays_o, rays_d,
                     xyz_min, xyz_max, (int)n_samples, total, rays_pts, mask_outbbox);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_sample_bg_pts_on_rays(const float *rays_o, const float *rays_d, const float *t_max,
                                           float bg_preserve, int64_t n_samples, int64_t n_rays,
                                           float *rays_pts, ugrid_stream_t s) {
  const int64_t total = n_samples * n_rays;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(k_sample_bg, dim3(ug_blocks(total, 256)), dim3(256), 0, ST(s), rays_o, rays_d, t_max,
                     bg_preserve, (int)n_samples, total, rays_pts);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_maskcache_lookup(const uint8_t *world, const float *xyz, const float *scale,
                                      const float *shift, int64_t sz_i, int64_t sz_j, int64_t sz_k,
                                      int64_t n_pts, uint8_t *out, ugrid_stream_t s) {
  if (n_pts <= 0) return 0;
  hipLaunchKernelGGL(k_maskcache, dim3(ug_blocks(n_pts, 256)), dim3(256), 0, ST(s), world, xyz, scale,
                     shift, sz_i, sz_j, sz_k, n_pts, out);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_raw2alpha(const float *density, float shift, float interval,
                               const float *interval_arr, int64_t n, float *exp_d, float *alpha,
                               ugrid_stream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_raw2alpha, dim3(ug_blocks(n, 256)), dim3(256), 0, ST(s), density, shift, interval,
                     interval_arr, n, exp_d, alpha);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_raw2alpha_backward(const float *exp_d, const float *grad_back, float interval,
                                        const float *interval_arr, int64_t n, float *grad,
                                        ugrid_stream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_raw2alpha_bwd, dim3(ug_blocks(n, 256)), dim3(256), 0, ST(s), exp_d, grad_back,
                     interval, interval_arr, n, grad);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_alpha2weight(const float *alpha, const int64_t *ray_id, int64_t n, int64_t n_rays,
                                  float *weight, float *T, float *alphainv_last, int64_t *i_start,
                                  int64_t *i_end, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  UG_HIP(hipMemsetAsync(i_start, 0, sizeof(int64_t) * n_rays, ST(s)));
  UG_HIP(hipMemsetAsync(i_end, 0, sizeof(int64_t) * n_rays, ST(s)));
  if (n > 0)
    hipLaunchKernelGGL(k_segments, dim3(ug_blocks(n, 256)), dim3(256), 0, ST(s), ray_id, n, i_start, i_end);
  // 4 rays (waves) per 256-thread block; empty rays just write alphainv_last = 1
  hipLaunchKernelGGL(k_alpha2weight, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), alpha,
                     n_rays, weight, T, alphainv_last, i_start, i_end);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_alpha2weight_backward(const float *alpha, const float *weight, const float *T,
                                           const float *alphainv_last, const int64_t *i_start,
                                           const int64_t *i_end, int64_t n, int64_t n_rays,
                                           const float *grad_weights, const float *grad_last,
                                           float *grad, ugrid_stream_t s) {
  if (n > 0) UG_HIP(hipMemsetAsync(grad, 0, sizeof(float) * n, ST(s)));
  if (n_rays <= 0 || n <= 0) return 0;
  hipLaunchKernelGGL(k_alpha2weight_bwd, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s),
                     alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last, grad);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_total_variation_add_grad(const float *param, float *grad, float wx, float wy,
                                              float wz, int dense_mode, int64_t sz_i, int64_t sz_j,
                                              int64_t sz_k, int64_t N, ugrid_stream_t s) {
  if (N <= 0) return 0;
  (void)wx;  // ignored by the reference as well (total_variation_kernel.cu:31-32)
  wy /= 6;
  wz /= 6;
  const bool vec = (sz_k % 4 == 0) && N < ((int64_t)1 << 31) && sz_i * sz_j * sz_k > 0 &&
                   ((((uintptr_t)param) | ((uintptr_t)grad)) & 15) == 0;
  if (vec) {
    const unsigned n4 = (unsigned)(N / 4);
    if (dense_mode && g_tv_xcd)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_vec4<true, 1>), dim3((n4 + 255) / 256), dim3(256), 0, ST(s), param, grad,
                         wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, n4);
    else if (dense_mode)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_vec4<true>), dim3((n4 + 255) / 256), dim3(256), 0, ST(s), param, grad,
                         wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, n4);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_vec4<false>), dim3((n4 + 255) / 256), dim3(256), 0, ST(s), param, grad,
                         wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, n4);
  } else if (dense_mode)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv<true>), dim3(ug_blocks(N, 256)), dim3(256), 0, ST(s), param,
                       grad, wy, wz, sz_i, sz_j, sz_k, N);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv<false>), dim3(ug_blocks(N, 256)), dim3(256), 0, ST(s), param,
                       grad, wy, wz, sz_i, sz_j, sz_k, N);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_cumdist_thres(const float *dist, float thres, int64_t n_rays, int64_t n_pts,
                                   uint8_t *mask, ugrid_stream_t s) {
  if (n_rays <= 0 || n_pts <= 0) return 0;
  hipLaunchKernelGGL(k_cumdist, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), dist, thres,
                     n_rays, n_pts, mask);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_segment_cumsum(const float *w, const float *s_, const int64_t *ray_id, int64_t n, int64_t n_rays,
                                    float *w_prefix, float *w_total, float *ws_prefix, float *ws_total,
                                    int64_t *seg_scratch, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  int64_t *i_start = seg_scratch, *i_end = seg_scratch + n_rays;   // zero = "ray without samples"
  UG_HIP(hipMemsetAsync(seg_scratch, 0, sizeof(int64_t) * 2 * n_rays, ST(s)));
  if (n > 0)
    hipLaunchKernelGGL(k_segments, dim3(ug_blocks(n, 256)), dim3(256), 0, ST(s), ray_id, n, i_start, i_end);
  hipLaunchKernelGGL(k_segment_cumsum, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), w, s_, n_rays,
                     i_start, i_end, w_prefix, w_total, ws_prefix, ws_total);
  UG_LAUNCH_CHECK();
  return 0;
}

// channel-last total_variation_add_grad: param / grad are [planes][sz_i][sz_j][sz_k][C] (C % 4 == 0, N < 2^31, 16-byte aligned)
static int ug_tv_cl(const float *param, float *grad, float wx, float wy, float wz, int dense_mode, int64_t sz_i, int64_t sz_j,
                    int64_t sz_k, int64_t C, int64_t N, const uint32_t *touch, ugrid_stream_t s);

extern "C" int ugrid_total_variation_add_grad_cl(const float *param, float *grad, float wx, float wy, float wz, int dense_mode,
                                                 int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t C, int64_t N,
                                                 ugrid_stream_t s) {
  return ug_tv_cl(param, grad, wx, wy, wz, dense_mode, sz_i, sz_j, sz_k, C, N, nullptr, s);
}

// masked mode with the touched-line bitmap of the gradient (ugrid_grid_query_backward_cl_touch): only marked lines are read
extern "C" int ugrid_total_variation_add_grad_cl_touch(const float *param, float *grad, float wx, float wy, float wz,
                                                       int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t C, int64_t N,
                                                       const uint32_t *touch, ugrid_stream_t s) {
  return ug_tv_cl(param, grad, wx, wy, wz, 0, sz_i, sz_j, sz_k, C, N, touch, s);
}

extern "C" int64_t ugrid_touch_words(int64_t N) { return ((N + 63) / 64 + 31) / 32; }

static int ug_tv_cl(const float *param, float *grad, float wx, float wy, float wz, int dense_mode, int64_t sz_i, int64_t sz_j,
                    int64_t sz_k, int64_t C, int64_t N, const uint32_t *touch, ugrid_stream_t s) {
  if (N <= 0) return 0;
  (void)wx;
  if (C % 4 != 0 || N >= ((int64_t)1 << 31) || ((((uintptr_t)param) | ((uintptr_t)grad)) & 15) != 0)
    return (int)hipErrorNotSupported;
  wy /= 6;
  wz /= 6;
  const unsigned n4 = (unsigned)(N / 4);
  if (dense_mode && g_tv_xcd)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<true, 0, 1>), dim3((n4 + 255) / 256), dim3(256), 0, ST(s), param, nullptr, grad,
                       nullptr, nullptr, wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, (int)C, n4, 0.f, 0.f, 0.f, 0.f, 0, (const uint32_t *)nullptr);
  else if (dense_mode)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<true, 0>), dim3((n4 + 255) / 256), dim3(256), 0, ST(s), param, nullptr, grad,
                       nullptr, nullptr, wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, (int)C, n4, 0.f, 0.f, 0.f, 0.f, 0, (const uint32_t *)nullptr);
  else if (touch) {
    const int64_t n_words = ugrid_touch_words(N);
    const int wpw = ug_touch_wpw(n_words);
    hipLaunchKernelGGL(k_tv_cl_touch, dim3(ug_blocks((n_words + wpw - 1) / wpw * UG_WAVE, 256)), dim3(256), 0, ST(s), param, grad, wy, wz, (int)sz_i,
                       (int)sz_j, (int)sz_k, (int)C, n4, touch, n_words, wpw);
  } else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<false, 0>), dim3((n4 + 255) / 256), dim3(256), 0, ST(s), param, nullptr, grad,
                       nullptr, nullptr, wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, (int)C, n4, 0.f, 0.f, 0.f, 0.f, 0, (const uint32_t *)nullptr);
  UG_LAUNCH_CHECK();
  return 0;
}

static int ug_tv_adam_dense_cl(const float *param, float *param_out, const float *grad, float *exp_avg, float *exp_avg_sq,
                               float wx, float wy, float wz, int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t C, int64_t N, int step,
                               float beta1, float beta2, float lr, float eps, int flags, uint32_t *touch, ugrid_stream_t s);

extern "C" int ugrid_tv_adam_dense_cl(const float *param, float *param_out, const float *grad, float *exp_avg,
                                      float *exp_avg_sq, float wx, float wy, float wz, int64_t sz_i, int64_t sz_j,
                                      int64_t sz_k, int64_t C, int64_t N, int step, float beta1, float beta2, float lr,
                                      float eps, int flags, ugrid_stream_t s) {
  return ug_tv_adam_dense_cl(param, param_out, grad, exp_avg, exp_avg_sq, wx, wy, wz, sz_i, sz_j, sz_k, C, N, step, beta1, beta2, lr,
                             eps, flags, nullptr, s);
}

// + the touched-line bitmap of the gradient: lines not marked are known to be zero and are not read; with the rezero flag
// the bitmap is cleared after the pass (the gradient is all zero again)
extern "C" int ugrid_tv_adam_dense_cl_touch(const float *param, float *param_out, const float *grad, float *exp_avg,
                                            float *exp_avg_sq, float wx, float wy, float wz, int64_t sz_i, int64_t sz_j,
                                            int64_t sz_k, int64_t C, int64_t N, int step, float beta1, float beta2, float lr,
                                            float eps, int flags, uint32_t *touch, ugrid_stream_t s) {
  return ug_tv_adam_dense_cl(param, param_out, grad, exp_avg, exp_avg_sq, wx, wy, wz, sz_i, sz_j, sz_k, C, N, step, beta1, beta2, lr,
                             eps, flags, touch, s);
}

static int ug_tv_adam_dense_cl(const float *param, float *param_out, const float *grad, float *exp_avg, float *exp_avg_sq,
                               float wx, float wy, float wz, int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t C, int64_t N, int step,
                               float beta1, float beta2, float lr, float eps, int flags, uint32_t *touch, ugrid_stream_t s) {
  if (N <= 0) return 0;
  const int skip_zero_grad = flags & 1, rezero = (flags >> 1) & 1;
  (void)wx;
  const uintptr_t al = (uintptr_t)param | (uintptr_t)param_out | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq;
  if (C % 4 != 0 || N >= ((int64_t)1 << 31) || (al & 15) != 0 || param == param_out) return (int)hipErrorNotSupported;
  wy /= 6;
  wz /= 6;
  const float step_size = lr * sqrtf(1 - powf(beta2, (float)step)) / (1 - powf(beta1, (float)step));
  const unsigned n4 = (unsigned)(N / 4);
  float *g = const_cast<float *>(grad);   // ADAM != 0 never writes the gradient
  const dim3 gr((n4 + 255) / 256), bl(256);
  // slab order (k_tv_cl_slab) when an i-plane is too large to stay in L2 across two plane-times: >= 512 KB per plane
  const int64_t row4 = sz_k * C / 4, plane_bytes = sz_j * row4 * 16;
  if (g_tv_xcd == 3 && sz_i >= 4 && sz_j >= 16 && plane_bytes >= (512 << 10) && N % (sz_i * sz_j * row4 * 4) == 0) {
    ug_tv_slab sl;
    sl.row4 = (unsigned)row4;
    // rows per slab: three slab-planes (+ the streams' working set) well inside the 4 MB L2 -> about 256 KB per slab-plane
    int64_t jw = (256 << 10) / (row4 * 16);
    jw = jw < 4 ? 4 : (jw > sz_j ? sz_j : jw);
    sl.jw = (unsigned)jw;
    sl.n_slab = (unsigned)((sz_j + jw - 1) / jw);
    sl.blocks_per_row_run = (unsigned)((jw * row4 + 255) / 256);
    const int64_t levels = N / (sz_i * sz_j * row4 * 4);
    const int64_t blocks = levels * sl.n_slab * sz_i * sl.blocks_per_row_run;
    if (blocks < ((int64_t)1 << 31)) {
#define UG_TV_SLAB_ARGS param, param_out, g, exp_avg, exp_avg_sq, wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, (int)C, n4, sl, step_size, beta1, beta2, eps, rezero, (const uint32_t *)touch
      if (skip_zero_grad) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_slab<1>), dim3((unsigned)blocks), bl, 0, ST(s), UG_TV_SLAB_ARGS);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_slab<2>), dim3((unsigned)blocks), bl, 0, ST(s), UG_TV_SLAB_ARGS);
#undef UG_TV_SLAB_ARGS
      UG_LAUNCH_CHECK();
      if (touch && rezero) UG_HIP(hipMemsetAsync(touch, 0, sizeof(uint32_t) * (size_t)ugrid_touch_words(N), ST(s)));
      return 0;
    }
  }
#define UG_TV_CL_ARGS param, param_out, g, exp_avg, exp_avg_sq, wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, (int)C, n4, step_size, beta1, beta2, eps, rezero, (const uint32_t *)touch
  if (g_tv_xcd >= 2) {
    if (skip_zero_grad) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<true, 1, 2>), gr, bl, 0, ST(s), UG_TV_CL_ARGS);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<true, 2, 2>), gr, bl, 0, ST(s), UG_TV_CL_ARGS);
  } else if (g_tv_xcd) {
    if (skip_zero_grad) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<true, 1, 1>), gr, bl, 0, ST(s), UG_TV_CL_ARGS);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<true, 2, 1>), gr, bl, 0, ST(s), UG_TV_CL_ARGS);
  } else {
    if (skip_zero_grad) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<true, 1>), gr, bl, 0, ST(s), UG_TV_CL_ARGS);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_cl_vec4<true, 2>), gr, bl, 0, ST(s), UG_TV_CL_ARGS);
  }
#undef UG_TV_CL_ARGS
  UG_LAUNCH_CHECK();
  if (touch && rezero) UG_HIP(hipMemsetAsync(touch, 0, sizeof(uint32_t) * (size_t)ugrid_touch_words(N), ST(s)));
  return 0;
}

extern "C" int ugrid_rays_of_a_view(int32_t H, int32_t W, const float *h_K9, const float *c2w, int inverse_y, int flip_x,
                                    int flip_y, int mode_center, const int64_t *pixel_index, int64_t n, float *rays_o,
                                    float *rays_d, float *viewdirs, ugrid_stream_t s) {
  if (n <= 0) return 0;
  if (H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
  ug_cam c;
  c.fx = h_K9[0]; c.fy = h_K9[4]; c.cx = h_K9[2]; c.cy = h_K9[5];
  c.W = W; c.H = H; c.inverse_y = inverse_y; c.flip_x = flip_x; c.flip_y = flip_y; c.center = mode_center;
  hipLaunchKernelGGL(k_rays_of_a_view, dim3(ug_blocks(n, 256)), dim3(256), 0, ST(s), c, c2w, pixel_index, n, rays_o, rays_d,
                     viewdirs);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_tv_adam_dense(const float *param, float *param_out, const float *grad, float *exp_avg,
                                   float *exp_avg_sq, float wx, float wy, float wz, int64_t sz_i, int64_t sz_j,
                                   int64_t sz_k, int64_t N, int step, float beta1, float beta2, float lr, float eps,
                                   int flags, ugrid_stream_t s) {
  if (N <= 0) return 0;
  const int skip_zero_grad = flags & 1, rezero = (flags >> 1) & 1;
  (void)wx;
  const uintptr_t al = (uintptr_t)param | (uintptr_t)param_out | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq;
  if (sz_k % 4 != 0 || N >= ((int64_t)1 << 31) || sz_i * sz_j * sz_k <= 0 || (al & 15) != 0 || param == param_out)
    return (int)hipErrorNotSupported;   // caller falls back to total_variation_add_grad + adam_upd
  wy /= 6;
  wz /= 6;
  const float step_size = lr * sqrtf(1 - powf(beta2, (float)step)) / (1 - powf(beta1, (float)step));
  const unsigned n4 = (unsigned)(N / 4);
  const dim3 gr((n4 + 255) / 256), bl(256);
#define UG_TV_ARGS param, param_out, grad, exp_avg, exp_avg_sq, wy, wz, (int)sz_i, (int)sz_j, (int)sz_k, n4, step_size, beta1, beta2, eps, rezero
  if (g_tv_xcd >= 2) {      // (canonical layout, C = 1: an i-plane is Y x Z floats -- 160 KB at G = 200 -- and stays in L2: no sweep needed)
    if (skip_zero_grad) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_adam_vec4<true, 2>), gr, bl, 0, ST(s), UG_TV_ARGS);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_adam_vec4<false, 2>), gr, bl, 0, ST(s), UG_TV_ARGS);
  } else if (g_tv_xcd) {
    if (skip_zero_grad) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_adam_vec4<true, 1>), gr, bl, 0, ST(s), UG_TV_ARGS);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_adam_vec4<false, 1>), gr, bl, 0, ST(s), UG_TV_ARGS);
  } else {
    if (skip_zero_grad) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_adam_vec4<true>), gr, bl, 0, ST(s), UG_TV_ARGS);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tv_adam_vec4<false>), gr, bl, 0, ST(s), UG_TV_ARGS);
  }
#undef UG_TV_ARGS
  UG_LAUNCH_CHECK();
  return 0;
}

// masked_adam_upd with the touched-line bitmap of a recycled gradient buffer: only marked lines are visited; the gradient
// comes back all zero and the bitmap cleared (mode 3 of ugrid_adam_upd restricted to the marked lines)
extern "C" int ugrid_masked_adam_upd_touch(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t N, int step,
                                           float beta1, float beta2, float lr, float eps, uint32_t *touch, ugrid_stream_t s) {
  if (N <= 0) return 0;
  if (!touch) return (int)hipErrorInvalidValue;
  const float step_size = lr * sqrtf(1 - powf(beta2, (float)step)) / (1 - powf(beta1, (float)step));
  const uintptr_t al = (uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq;
  if ((al & 15) != 0) return (int)hipErrorNotSupported;
  const int64_t n4 = N / 4, n_words = ugrid_touch_words(N);
  // one word per wave for the masked Adam: its body is one load and (where the gradient is non-zero) three more -- the many short waves
  // hide that latency better than a few waves walking 16 words each (measured: 0.24 against 0.48 ms on S3's k0 grid, visit V); the TV
  // body's eight loads per element like the longer walk (0.88 -> 0.74 ms)
  const int wpw = 1;
  if (n4 > 0)
    hipLaunchKernelGGL(k_adam_vec4_touch, dim3(ug_blocks((n_words + wpw - 1) / wpw * UG_WAVE, 256)), dim3(256), 0, ST(s), (float4 *)param,
                       (const float4 *)grad, (float4 *)exp_avg, (float4 *)exp_avg_sq, n4, step_size, beta1, beta2, eps, touch, n_words, wpw);
  if (n4 * 4 < N)      // the last 1-3 elements, whatever their line says
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_adam_scalar<1, true>), dim3(1), dim3(256), 0, ST(s), param, grad, exp_avg, exp_avg_sq,
                       (const float *)nullptr, n4 * 4, N, step_size, beta1, beta2, eps);
  UG_LAUNCH_CHECK();
  UG_HIP(hipMemsetAsync(touch, 0, sizeof(uint32_t) * (size_t)n_words, ST(s)));
  return 0;
}

// ---- multi-tensor Adam (round 5): the small parameters of a model (the rgbnet's six tensors: 22 k elements) in ONE launch instead
// of one launch -- and one host round trip through the binding -- each (masked_adam.py:43-75 loops over the parameters; a DVGO
// training step spent 0.25 ms of its 1.3 ms issuing eight such updates).  Element for element the arithmetic of ugrid_adam_upd
// (ug_adam_one), so the results are bit-identical to the per-tensor calls.
#define UG_ADAM_MULTI_MAX 16
struct ug_adam_table {
  float *param[UG_ADAM_MULTI_MAX];
  const float *grad[UG_ADAM_MULTI_MAX];
  float *m[UG_ADAM_MULTI_MAX], *v[UG_ADAM_MULTI_MAX];
  float step_size[UG_ADAM_MULTI_MAX];
  int32_t first_block[UG_ADAM_MULTI_MAX + 1];      // blocks [first_block[t], first_block[t+1]) work on tensor t
  int64_t numel[UG_ADAM_MULTI_MAX];
  int32_t n;
};

template <int MODE>
__global__ void __launch_bounds__(256) k_adam_multi(ug_adam_table tab, float beta1, float beta2, float eps) {
  int t = 0;
  while (t + 1 < tab.n && (int)blockIdx.x >= tab.first_block[t + 1]) ++t;     // wave-uniform, <= 16 steps
  const int64_t i = (int64_t)((int)blockIdx.x - tab.first_block[t]) * 256 + threadIdx.x;
  if (i >= tab.numel[t]) return;
  const float g = tab.grad[t][i];
  if (MODE == 1 && !(g != 0.f)) return;
  float p = tab.param[t][i], m = tab.m[t][i], v = tab.v[t][i];
  ug_adam_one<MODE>(p, g, m, v, 1.f, tab.step_size[t], beta1, beta2, eps);
  tab.param[t][i] = p;
  tab.m[t][i] = m;
  tab.v[t][i] = v;
}

extern "C" int ugrid_adam_upd_multi(const ugrid_adam_item *items, int32_t n_items, float beta1, float beta2, float eps,
                                    int32_t mode, ugrid_stream_t s) {
  if (n_items <= 0) return 0;
  if (!items || (mode != 0 && mode != 1)) return (int)hipErrorInvalidValue;
  for (int32_t base = 0; base < n_items; base += UG_ADAM_MULTI_MAX) {
    ug_adam_table tab;
    tab.n = 0;
    int64_t blocks = 0;
    for (int32_t k = base; k < n_items && tab.n < UG_ADAM_MULTI_MAX; ++k) {
      const ugrid_adam_item &it = items[k];
      if (it.numel <= 0) continue;
      if (!it.param || !it.grad || !it.exp_avg || !it.exp_avg_sq || it.numel > ((int64_t)1 << 30)) return (int)hipErrorInvalidValue;
      const int t = tab.n++;
      tab.param[t] = it.param; tab.grad[t] = it.grad; tab.m[t] = it.exp_avg; tab.v[t] = it.exp_avg_sq;
      tab.numel[t] = it.numel;
      // host-side, in float, like the reference (adam_upd_kernel.cu:72) and ugrid_adam_upd
      tab.step_size[t] = it.lr * sqrtf(1 - powf(beta2, (float)it.step)) / (1 - powf(beta1, (float)it.step));
      tab.first_block[t] = (int32_t)blocks;
      blocks += (it.numel + 255) / 256;
    }
    if (tab.n == 0) continue;
    tab.first_block[tab.n] = (int32_t)blocks;
    if (mode == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_adam_multi<0>), dim3((unsigned)blocks), dim3(256), 0, ST(s), tab, beta1, beta2, eps);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_adam_multi<1>), dim3((unsigned)blocks), dim3(256), 0, ST(s), tab, beta1, beta2, eps);
    UG_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int ugrid_adam_upd(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                              const float *perlr, int64_t N, int step, float beta1, float beta2, float lr,
                              float eps, int mode, ugrid_stream_t s) {
  if (N <= 0) return 0;
  // host-side, in float, like the reference (adam_upd_kernel.cu:72)
  const float step_size = lr * sqrtf(1 - powf(beta2, (float)step)) / (1 - powf(beta1, (float)step));
  switch (mode) {
    case 0: return ug_adam_launch<0>(param, grad, exp_avg, exp_avg_sq, nullptr, N, step_size, beta1, beta2, eps, ST(s));
    case 1: return ug_adam_launch<1>(param, grad, exp_avg, exp_avg_sq, nullptr, N, step_size, beta1, beta2, eps, ST(s));
    case 2:
      if (!perlr) return (int)hipErrorInvalidValue;
      return ug_adam_launch<2>(param, grad, exp_avg, exp_avg_sq, perlr, N, step_size, beta1, beta2, eps, ST(s));
    case 3: return ug_adam_launch<1, true>(param, grad, exp_avg, exp_avg_sq, nullptr, N, step_size, beta1, beta2, eps, ST(s));
    default: return (int)hipErrorInvalidValue;
  }
}
