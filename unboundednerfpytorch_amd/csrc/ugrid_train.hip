// Training tail in two kernels per direction: sigmoid + per-ray compositing + the loss terms of run_train.py:254-279
// (main MSE, the image-space Fourier MSE, entropy_last, nearclip, flatten_eff_distloss, rgbper) and their hand-written derivatives.
//
// The composed torch chain this replaces (FourierGrid_model.py:636-672 after the rgbnet, run_train.py:254-279) is ~45
// launches forward and ~90 backward on [M] / [M,3] / [R,3] arrays that together hold a few MB: launch-bound.  Here one
// wave owns one ray (ray_id is ascending: the model's compaction is ray-major), lane = sample in 64-sample rounds.
//
//   per sample i of ray r :  rgb_i = sigmoid(logit_i)                                   (FourierGrid_model.py:636)
//   rgb_marched[r]        =  sum_i w_i rgb_i + alphainv_last[r] * bg[r]                 (:638-647, bg = rand_bkgd draw or none)
//   mse                   =  mean_{r,c} (rgb_marched - target)^2                        (run_train.py:254)
//   freq                  =  mean_{r,k} (Re FFT_3(rgb_marched)_k - Re FFT_3(target)_k)^2   (run_train.py:255, FourierMSELoss
//                            FourierGrid_model.py:112-129: the FFT runs over the COLOUR axis, n = 3, only the real part is used).
//                            Re FFT_3 is linear: with e = rgb_marched - target its three outputs are f0 = e0 + e1 + e2 and, twice,
//                            f1 = e0 - (e1 + e2) / 2  (cos(2 pi / 3) = cos(4 pi / 3) = -1/2)  ->  freq = mean_r (f0^2 + 2 f1^2) / 3
//   entropy_last          =  mean_r -(p log p + (1-p) log(1-p)),  p = clamp(alphainv_last, 1e-6, 1-1e-6)   (:258-261)
//   nearclip              =  sum_{i: t_i < near} (density_i - stop_grad(density_i))      (:262-265; value 0, gradient 1)
//   distortion            =  1/n_d sum_i [ 2 w_i (s_i W_<i - WS_<i) + w_i^2 interval / 3 ],  n_d = ray_id.max() + 1   (:274)
//   rgbper                =  1/n_rays sum_i stop_grad(w_i) sum_c (rgb_i - target[r])^2    (:276-278)
//   loss = w_main mse + w_freq freq + w_ent entropy_last + w_near nearclip + w_dist distortion + w_per rgbper
//
// Per-ray partial sums go to a [R,5] array that ONE block reduces in a fixed order (deterministic loss value); the
// exclusive running sums W_<i, WS_<i are formed sequentially along the ray exactly as k_segment_cumsum forms them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugrid_common.h"
#include "ugrid_hip.h"

#define ST(s) ((hipStream_t)(s))

struct ug_loss_coef {
  float w_main, w_ent, w_dist, w_per, w_near, near_thres, interval, n_rays, w_freq;
};
#define UG_LOSS_PARTIALS 5

__device__ __forceinline__ float ug_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, UG_WAVE);
  return v;
}

__device__ __forceinline__ float ug_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// s = 1 - 1 / (1 + t) (FourierGrid_model.py:649) when the caller passes no s array
__device__ __forceinline__ float ug_s_at(const float *__restrict__ s, const float *__restrict__ t, int64_t i) {
  return s ? s[i] : 1.0f - 1.0f / (1.0f + t[i]);
}

// one wave per ray
__global__ void __launch_bounds__(256)
k_render_loss_fwd(const float *__restrict__ logits, const float *__restrict__ weights, const float *__restrict__ s,
                  const float *__restrict__ t, const float *__restrict__ ainv, const float *__restrict__ bg, const float *__restrict__ target,
                  const int64_t *__restrict__ i_start, const int64_t *__restrict__ i_end, int64_t n_rays, ug_loss_coef c,
                  float *__restrict__ rgb_marched, float *__restrict__ ray_tot, float *__restrict__ partial) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= n_rays) return;
  const int lane = ug_lane();
  const int64_t i_s = i_start[r], i_e = i_end[r];
  const float t0 = target[3 * r], t1 = target[3 * r + 1], t2 = target[3 * r + 2];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, per = 0.f, dist = 0.f, cw = 0.f, cws = 0.f;
  for (int64_t base = i_s; base < i_e; base += UG_WAVE) {
    const int64_t i = base + lane;
    const bool on = i < i_e;
    const float w = on ? weights[i] : 0.f;
    const float si = on ? ug_s_at(s, t, i) : 0.f;
    const float ws = w * si;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    if (on) {
      r0 = ug_sigmoid(logits[3 * i]);
      r1 = ug_sigmoid(logits[3 * i + 1]);
      r2 = ug_sigmoid(logits[3 * i + 2]);
    }
    const int cnt = (int)((i_e - base) < UG_WAVE ? (i_e - base) : UG_WAVE);
    float w_pre = 0.f, ws_pre = 0.f;
    for (int k = 0; k < cnt; ++k) {
      if (lane == k) { w_pre = cw; ws_pre = cws; }
      cw = cw + ug_readlane_f(w, k);
      cws = cws + ug_readlane_f(ws, k);
    }
    if (on) {
      a0 += w * r0;
      a1 += w * r1;
      a2 += w * r2;
      const float d0 = r0 - t0, d1 = r1 - t1, d2 = r2 - t2;
      per += (d0 * d0 + d1 * d1 + d2 * d2) * w;
      dist += 2.f * w * (si * w_pre - ws_pre) + (1.f / 3.f) * c.interval * (w * w);
    }
  }
  a0 = ug_wave_sum(a0);
  a1 = ug_wave_sum(a1);
  a2 = ug_wave_sum(a2);
  per = ug_wave_sum(per);
  dist = ug_wave_sum(dist);
  if (lane == 0) {
    const float av = ainv[r];
    if (bg) {
      a0 += av * bg[3 * r];
      a1 += av * bg[3 * r + 1];
      a2 += av * bg[3 * r + 2];
    }
    rgb_marched[3 * r] = a0;
    rgb_marched[3 * r + 1] = a1;
    rgb_marched[3 * r + 2] = a2;
    ray_tot[2 * r] = cw;
    ray_tot[2 * r + 1] = cws;
    const float e0 = a0 - t0, e1 = a1 - t1, e2 = a2 - t2;
    const float p = fminf(fmaxf(av, 1e-6f), 1.f - 1e-6f);
    const float f0 = (e0 + e1) + e2, f1 = e0 - 0.5f * (e1 + e2);
    partial[UG_LOSS_PARTIALS * r] = e0 * e0 + e1 * e1 + e2 * e2;
    partial[UG_LOSS_PARTIALS * r + 1] = -(p * logf(p) + (1.f - p) * logf(1.f - p));
    partial[UG_LOSS_PARTIALS * r + 2] = per;
    partial[UG_LOSS_PARTIALS * r + 3] = dist;
    partial[UG_LOSS_PARTIALS * r + 4] = f0 * f0 + 2.f * (f1 * f1);
  }
}

// one block: fixed-order reduction of the [R,5] partials, then the weighted sum.  out = {loss, mse}
__global__ void __launch_bounds__(256)
k_render_loss_final(const float *__restrict__ partial, int64_t n_rays, const int64_t *__restrict__ ray_id, int64_t n,
                    ug_loss_coef c, float *__restrict__ out, const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(n, n_dev);
  constexpr int NP = UG_LOSS_PARTIALS;
  __shared__ float red[NP][256];
  float acc[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) acc[k] = 0.f;
  for (int64_t r = threadIdx.x; r < n_rays; r += 256)
#pragma unroll
    for (int k = 0; k < NP; ++k) acc[k] += partial[NP * r + k];
#pragma unroll
  for (int k = 0; k < NP; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int k = 0; k < NP; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float R = (float)n_rays;
    const float mse = red[0][0] / (3.f * R);
    float loss = c.w_main * mse;
    if (c.w_freq != 0.f) loss += c.w_freq * (red[4][0] / (3.f * R));
    if (c.w_ent > 0.f) loss += c.w_ent * (red[1][0] / R);
    if (c.w_dist > 0.f && n > 0) loss += c.w_dist * (red[3][0] / (float)(ray_id[n - 1] + 1));
    if (c.w_per > 0.f) loss += c.w_per * (red[2][0] / c.n_rays);
    out[0] = loss;
    out[1] = mse;
  }
}

__global__ void __launch_bounds__(256)
k_render_loss_bwd(const float *__restrict__ logits, const float *__restrict__ weights, const float *__restrict__ s,
                  const float *__restrict__ t, const float *__restrict__ ainv, const float *__restrict__ bg,
                  const float *__restrict__ target, const int64_t *__restrict__ ray_id, int64_t n,
                  const int64_t *__restrict__ i_start, const int64_t *__restrict__ i_end, int64_t n_rays, ug_loss_coef c,
                  const float *__restrict__ rgb_marched, const float *__restrict__ ray_tot,
                  const float *__restrict__ grad_loss, float *__restrict__ g_logits, float *__restrict__ g_weights,
                  float *__restrict__ g_ainv, float *__restrict__ g_density, const int64_t *__restrict__ n_dev) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= n_rays) return;
  UG_DEVN_CLAMP(n, n_dev);
  const int lane = ug_lane();
  const int64_t i_s = i_start[r], i_e = i_end[r];
  const float g = grad_loss[0];
  const float R = (float)n_rays;
  const float t0 = target[3 * r], t1 = target[3 * r + 1], t2 = target[3 * r + 2];
  const float k_mse = g * c.w_main * 2.f / (3.f * R);
  float m0 = k_mse * (rgb_marched[3 * r] - t0), m1 = k_mse * (rgb_marched[3 * r + 1] - t1), m2 = k_mse * (rgb_marched[3 * r + 2] - t2);
  if (c.w_freq != 0.f) {
    // d freq / d e = 2 / (3 R) [ f0 (1,1,1) + 2 f1 (1,-1/2,-1/2) ]: the transposed 3 x 3 map applied to (f0, f1, f1)
    const float e0 = rgb_marched[3 * r] - t0, e1 = rgb_marched[3 * r + 1] - t1, e2 = rgb_marched[3 * r + 2] - t2;
    const float f0 = (e0 + e1) + e2, f1 = e0 - 0.5f * (e1 + e2);
    const float k_freq = g * c.w_freq * 2.f / (3.f * R);
    m0 += k_freq * (f0 + 2.f * f1);
    m1 += k_freq * (f0 - f1);
    m2 += k_freq * (f0 - f1);
  }
  const float k_per = c.w_per > 0.f ? g * c.w_per * 2.f / c.n_rays : 0.f;
  const float k_dist = (c.w_dist > 0.f && n > 0) ? g * c.w_dist / (float)(ray_id[n - 1] + 1) : 0.f;
  const float k_near = g * c.w_near;
  const float w_tot = ray_tot[2 * r], ws_tot = ray_tot[2 * r + 1];
  float cw = 0.f, cws = 0.f;
  for (int64_t base = i_s; base < i_e; base += UG_WAVE) {
    const int64_t i = base + lane;
    const bool on = i < i_e;
    const float w = on ? weights[i] : 0.f;
    const float si = on ? ug_s_at(s, t, i) : 0.f;
    const float ws = w * si;
    const int cnt = (int)((i_e - base) < UG_WAVE ? (i_e - base) : UG_WAVE);
    float w_pre = 0.f, ws_pre = 0.f;
    for (int k = 0; k < cnt; ++k) {
      if (lane == k) { w_pre = cw; ws_pre = cws; }
      cw = cw + ug_readlane_f(w, k);
      cws = cws + ug_readlane_f(ws, k);
    }
    if (on) {
      const float r0 = ug_sigmoid(logits[3 * i]), r1 = ug_sigmoid(logits[3 * i + 1]), r2 = ug_sigmoid(logits[3 * i + 2]);
      const float d0 = m0 * w + k_per * (r0 - t0) * w, d1 = m1 * w + k_per * (r1 - t1) * w,
                  d2 = m2 * w + k_per * (r2 - t2) * w;
      g_logits[3 * i] = d0 * ((1.f - r0) * r0);
      g_logits[3 * i + 1] = d1 * ((1.f - r1) * r1);
      g_logits[3 * i + 2] = d2 * ((1.f - r2) * r2);
      const float w_after = w_tot - (w_pre + w), ws_after = ws_tot - (ws_pre + ws);
      const float d_pair = 2.f * (si * (w_pre - w_after) + (ws_after - ws_pre));
      const float d_self = (1.f / 3.f) * c.interval * 2.f * w;
      g_weights[i] = (m0 * r0 + m1 * r1 + m2 * r2) + k_dist * (d_pair + d_self);
      g_density[i] = (c.w_near > 0.f && t[i] < c.near_thres) ? k_near : 0.f;
    }
  }
  if (lane == 0) {
    const float av = ainv[r];
    float ga = 0.f;
    if (bg) ga = m0 * bg[3 * r] + m1 * bg[3 * r + 1] + m2 * bg[3 * r + 2];
    if (c.w_ent > 0.f && av >= 1e-6f && av <= 1.f - 1e-6f) ga += g * c.w_ent / R * -(logf(av) - logf(1.f - av));
    g_ainv[r] = ga;
  }
}

static ug_loss_coef ug_coef(const float *h) {
  ug_loss_coef c;
  c.w_main = h[0]; c.w_ent = h[1]; c.w_dist = h[2]; c.w_per = h[3]; c.w_near = h[4]; c.near_thres = h[5]; c.interval = h[6];
  c.n_rays = h[7];
  c.w_freq = h[8];
  return c;
}

__global__ void k_zero_i64(int64_t *__restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0;
}

// k_segments of ugrid_ops.hip (ray_id ascending -> [i_start, i_end) per ray, empty rays 0,0)
// (n_dev: the sample count on the device, ug_devn in ugrid_common.h; null = n)
__global__ void k_loss_segments(const int64_t *__restrict__ ray_id, int64_t n, int64_t *__restrict__ i_start,
                                int64_t *__restrict__ i_end, const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(n, n_dev);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
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
}

extern "C" int ugrid_render_loss(const float *logits, const float *weights, const float *s, const float *t, const float *alphainv_last,
                                 const float *bg, const float *target, const int64_t *ray_id, int64_t n, int64_t n_rays,
                                 const float *h_coef9, int64_t *seg_scratch, float *rgb_marched, float *ray_tot,
                                 float *partial, float *out2, ugrid_stream_t st) {
  if (n_rays <= 0 || (n > 0 && !s && !t)) return (int)hipErrorInvalidValue;      // (no samples: empty arrays have no address)
  int64_t *i_start = seg_scratch, *i_end = seg_scratch + n_rays;
  // (a kernel, not hipMemsetAsync: the memset NODE of a replayed hipGraph does not invalidate what the previous replay's kernels left in
  // L2 -- DESIGN.md 4.2, the render path's tile counters -- and the sync-free training step is captured with this call inside)
  hipLaunchKernelGGL(k_zero_i64, dim3(ug_blocks(2 * n_rays, 256) < 1024 ? ug_blocks(2 * n_rays, 256) : 1024), dim3(256), 0, ST(st), seg_scratch,
                     2 * n_rays);
  if (n > 0) hipLaunchKernelGGL(k_loss_segments, dim3(ug_blocks(ug_launch_rows(n), 256)), dim3(256), 0, ST(st), ray_id, n, i_start, i_end, ug_tl_devn.ptr);
  const ug_loss_coef c = ug_coef(h_coef9);
  hipLaunchKernelGGL(k_render_loss_fwd, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(st), logits, weights, s, t,
                     alphainv_last, bg, target, i_start, i_end, n_rays, c, rgb_marched, ray_tot, partial);
  hipLaunchKernelGGL(k_render_loss_final, dim3(1), dim3(256), 0, ST(st), partial, n_rays, ray_id, n, c, out2, ug_tl_devn.ptr);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_render_loss_backward(const float *logits, const float *weights, const float *s, const float *t,
                                          const float *alphainv_last, const float *bg, const float *target,
                                          const int64_t *ray_id, int64_t n, int64_t n_rays, const float *h_coef9,
                                          const int64_t *seg_scratch, const float *rgb_marched, const float *ray_tot,
                                          const float *grad_loss, float *g_logits, float *g_weights, float *g_alphainv_last,
                                          float *g_density, ugrid_stream_t st) {
  if (n_rays <= 0) return (int)hipErrorInvalidValue;
  const ug_loss_coef c = ug_coef(h_coef9);
  hipLaunchKernelGGL(k_render_loss_bwd, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(st), logits, weights, s, t,
                     alphainv_last, bg, target, ray_id, n, seg_scratch, seg_scratch + n_rays, n_rays, c, rgb_marched, ray_tot,
                     grad_loss, g_logits, g_weights, g_alphainv_last, g_density, ug_tl_devn.ptr);
  UG_LAUNCH_CHECK();
  return 0;
}

// Rows of the rgbnet's input, [k0 | viewdir | sin(viewdir 2^k) | cos(viewdir 2^k)] (FourierGrid_model.py:631-635: the view embedding
// is formed per ray, indexed by ray_id and concatenated behind the k0 features -- six elementwise launches and two concatenations
// there), one thread per output element so the stores coalesce.  Column order inside the sin / cos groups is torch's
// (viewdirs.unsqueeze(-1) * viewfreq).flatten(-2): axis-major, frequency-minor.
template <typename IDX>      // uint32_t when the element count fits (a 64-bit division is ~100 VALU instructions, the kernel's largest cost)
__global__ void __launch_bounds__(256)
k_rgbnet_features(const float *__restrict__ k0, int C, const float *__restrict__ viewdirs, const float *__restrict__ freq, int pe,
                  const int64_t *__restrict__ ray_id, int64_t total, float *__restrict__ out, const int64_t *__restrict__ n_dev) {
  const int K = C + 3 + 6 * pe;
  if (n_dev) {                                   // rows on the device (ug_devn): total = rows * K
    const int64_t td = *n_dev * K;
    if (td < total) total = td;
  }
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = (int64_t)((IDX)idx / (IDX)K);
    const int j = (int)(idx - m * K);
    if (j < C) {
      out[idx] = k0[m * C + j];
      continue;
    }
    const float *v = viewdirs + 3 * (ray_id ? ray_id[m] : m);
    int e = j - C;
    if (e < 3) {
      out[idx] = v[e];
      continue;
    }
    e -= 3;
    const bool is_cos = e >= 3 * pe;
    if (is_cos) e -= 3 * pe;
    const int a = e / pe;
    const float x = v[a] * freq[e - a * pe];
    out[idx] = is_cos ? cosf(x) : sinf(x);
  }
}

// out[m] = [k0[m] | ray_rows[ray_id[m]]]: the view embedding formed once per RAY (k_rgbnet_features over the rays) and gathered --
// a ray's ~15 surviving samples share its 24 sines and cosines, which were most of the one-pass kernel's time
template <typename IDX>
__global__ void __launch_bounds__(256)
k_rgbnet_rows(const float *__restrict__ k0, int C, const float *__restrict__ ray_rows, int E, const int64_t *__restrict__ ray_id,
              int64_t total, float *__restrict__ out, const int64_t *__restrict__ n_dev) {
  const int K = C + E;
  if (n_dev) {
    const int64_t td = *n_dev * K;
    if (td < total) total = td;
  }
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = (int64_t)((IDX)idx / (IDX)K);
    const int j = (int)(idx - m * K);
    out[idx] = j < C ? k0[m * C + j] : ray_rows[ray_id[m] * E + (j - C)];
  }
}

extern "C" int ugrid_rgbnet_features(const float *k0, int32_t n_k0, const float *viewdirs, int64_t n_rays, const float *viewfreq, int32_t pe,
                                     const int64_t *ray_id, int64_t m, float *ray_rows, float *out, ugrid_stream_t st) {
  if (n_k0 < 0 || pe < 0 || m < 0 || n_rays < 0) return (int)hipErrorInvalidValue;
  const int E = 3 + 6 * pe;
  const int64_t total = m * (n_k0 + E);
  if (total == 0) return 0;                                  // (no samples: empty arrays have no address)
  if ((n_k0 > 0 && !k0) || (pe > 0 && !viewfreq) || !viewdirs || !out) return (int)hipErrorInvalidValue;
#define UG_FEAT(KERNEL, N, NMAX, ...)      /* N: elements the grid is sized for; NMAX: the largest element index + 1 */           \
  if ((NMAX) < ((int64_t)1 << 32))                                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<uint32_t>), dim3(ug_blocks((N), 256)), dim3(256), 0, ST(st), __VA_ARGS__);            \
  else                                                                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<uint64_t>), dim3(ug_blocks((N), 256)), dim3(256), 0, ST(st), __VA_ARGS__);
  if (ray_id && ray_rows && n_rays > 0 && m >= 2 * n_rays) {
    const int64_t per_ray = n_rays * E;
    const int64_t total_l = ug_launch_rows(m) * (n_k0 + E);
    UG_FEAT(k_rgbnet_features, per_ray, per_ray, nullptr, 0, viewdirs, viewfreq, pe, nullptr, per_ray, ray_rows, (const int64_t *)nullptr)
    UG_FEAT(k_rgbnet_rows, total_l, total, k0, n_k0, ray_rows, E, ray_id, total, out, ug_tl_devn.ptr)
  } else {
    const int64_t total_l = ug_launch_rows(m) * (n_k0 + E);
    UG_FEAT(k_rgbnet_features, total_l, total, k0, n_k0, viewdirs, viewfreq, pe, ray_id, total, out, ug_tl_devn.ptr)
  }
#undef UG_FEAT
  UG_LAUNCH_CHECK();
  return 0;
}
