// fp64 twins of the drop-in ops (include/ugrid_hip_f64.h) -- built into libugrid_hip_f64.so, a library of its own.
//
// The reference's four extension modules dispatch on the tensors' type (AT_DISPATCH_FLOATING_TYPES: float and double).  No caller
// on the rendering / training path passes doubles (the models are fp32), so these kernels are NOT hot: one lane per item, plain
// loops, nothing tuned -- what matters is that a double tensor gets the reference's ARITHMETIC, which is not "everything in
// double".  The reference's kernels keep many intermediates in `float` whatever scalar_t is (ray parameters, the transmittance
// T_cum, the TV accumulator, the running distance, the sampling positions) and take their scalar arguments as `const float`;
// a product `float * double` is a double, `float += double` rounds back to float.  Each kernel below states those types where
// the reference's algorithm fixes them (file:line in the header) and is pinned, bit for bit, on the reference's own kernels
// compiled for gfx950 and called with double tensors (tests/test_gpu_ref_native.py, oracle/_ref).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugrid_hip_f64.h"

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(int64_t n) { return (unsigned)((n + kThreads - 1) / kThreads); }
inline hipStream_t S(ugrid_stream_t s) { return (hipStream_t)s; }
inline int launched() { return (int)hipGetLastError(); }

__device__ __forceinline__ int64_t gid() { return (int64_t)blockIdx.x * blockDim.x + threadIdx.x; }

// ---- rays against the box -----------------------------------------------------------------------------------------------
struct RaySpan { float t0, t1; };

// render_utils_kernel.cu:22-34: the direction components and the six slab distances are FLOAT variables; the subtraction and
// the division run in double (double operands) and are rounded when stored
__device__ __forceinline__ RaySpan ray_span(const double *o, const double *d, const double *lo, const double *hi, float near, float far) {
  float v[3], a[3], b[3];
  for (int c = 0; c < 3; ++c) {
    v[c] = (float)((d[c] == 0.0) ? 1e-6 : d[c]);
    a[c] = (float)((hi[c] - o[c]) / (double)v[c]);
    b[c] = (float)((lo[c] - o[c]) / (double)v[c]);
  }
  RaySpan r;
  r.t0 = fmaxf(fminf(fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(a[2], b[2])), far), near);
  r.t1 = fmaxf(fminf(fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(a[2], b[2])), far), near);
  return r;
}

// render_utils_kernel.cu:48-51, 70-73: `const float rnorm = sqrt(double sum)`
__device__ __forceinline__ float dir_norm(const double *d) { return (float)sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); }

// render_utils_kernel.cu:53: max(ceil((t_max - t_min) * rnorm / stepdist), 1.) in double
__device__ __forceinline__ int64_t step_count(double t0, double t1, float rnorm, float stepdist) {
  const double c = ceil((t1 - t0) * (double)rnorm / (double)stepdist);
  return (int64_t)(c > 1. ? c : 1.);
}

__global__ void k_span(const double *o, const double *d, const double *lo, const double *hi, float near, float far, int64_t n, double *t0,
                       double *t1) {
  const int64_t r = gid();
  if (r >= n) return;
  const RaySpan s = ray_span(o + 3 * r, d + 3 * r, lo, hi, near, far);
  t0[r] = (double)s.t0;
  t1[r] = (double)s.t1;
}

__global__ void k_steps(const double *d, const double *t0, const double *t1, float stepdist, int64_t n, int64_t *out) {
  const int64_t r = gid();
  if (r >= n) return;
  out[r] = step_count(t0[r], t1[r], dir_norm(d + 3 * r), stepdist);
}

__global__ void k_start_dir(const double *o, const double *d, const double *t0, int64_t n, double *start, double *dir) {
  const int64_t r = gid();
  if (r >= n) return;
  const float rn = dir_norm(d + 3 * r);
  for (int c = 0; c < 3; ++c) {
    start[3 * r + c] = o[3 * r + c] + d[3 * r + c] * t0[r];
    dir[3 * r + c] = d[3 * r + c] / (double)rn;
  }
}

__global__ void k_span_steps(const double *o, const double *d, const double *lo, const double *hi, float near, float far, float stepdist,
                             int64_t n, double *t0, double *t1, int64_t *steps) {
  const int64_t r = gid();
  if (r >= n) return;
  const RaySpan s = ray_span(o + 3 * r, d + 3 * r, lo, hi, near, far);
  t0[r] = (double)s.t0;
  t1[r] = (double)s.t1;
  steps[r] = step_count((double)s.t0, (double)s.t1, dir_norm(d + 3 * r), stepdist);
}

// inclusive prefix sums of n int64 counts by ONE workgroup (ray counts are at most a few million; this path is not hot)
__global__ void __launch_bounds__(1024) k_prefix(const int64_t *in, int64_t n, int64_t *out, int64_t *total) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + 1023) / 1024, lo = t * per, hi = lo + per < n ? lo + per : n;
  int64_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int64_t run = 0;
    for (int k = 0; k < 1024; ++k) {
      const int64_t v = part[k];
      part[k] = run;
      run += v;
    }
    *total = run;
  }
  __syncthreads();
  int64_t run = part[t];
  for (int64_t i = lo; i < hi; ++i) {
    run += in[i];
    out[i] = run;
  }
}

// render_utils_kernel.cu:165-191: `dist`, px, py, pz are FLOAT; the start point and unit direction are the double arrays of
// infer_ray_start_dir.  The owner ray of a sample is found by bisection in the prefix sums (the reference builds ray_id / step_id
// with two more scans).
__global__ void k_fill(const double *o, const double *d, const double *lo, const double *hi, const double *t0, const int64_t *cum, float stepdist,
                       int64_t n_rays, int64_t total, double *pts, uint8_t *outside, int64_t *ray_id, int64_t *step_id) {
  const int64_t i = gid();
  if (i >= total) return;
  int64_t a = 0, b = n_rays - 1;
  while (a < b) {
    const int64_t m = (a + b) >> 1;
    if (cum[m] > i) b = m; else a = m + 1;
  }
  const int64_t r = a, s = i - (r ? cum[r - 1] : 0);
  const float rn = dir_norm(d + 3 * r);
  const float dist = stepdist * (float)(int)s;
  float p[3];
  for (int c = 0; c < 3; ++c) {
    const double start = o[3 * r + c] + d[3 * r + c] * t0[r];
    const double dir = d[3 * r + c] / (double)rn;
    p[c] = (float)(start + dir * (double)dist);
    pts[3 * i + c] = (double)p[c];
  }
  outside[i] = (uint8_t)((lo[0] > (double)p[0]) | (lo[1] > (double)p[1]) | (lo[2] > (double)p[2]) | (hi[0] < (double)p[0]) |
                         (hi[1] < (double)p[1]) | (hi[2] < (double)p[2]));
  ray_id[i] = r;
  step_id[i] = s;
}

// render_utils_kernel.cu:260-268
__global__ void k_ndc(const double *o, const double *d, const double *lo, const double *hi, int n_samples, int64_t total, double *pts,
                      uint8_t *outside) {
  const int64_t i = gid();
  if (i >= total) return;
  const int64_t r = i / n_samples;
  const int s = (int)(i - r * n_samples);
  const float dist = ((float)s) / (float)(n_samples - 1);
  float p[3];
  for (int c = 0; c < 3; ++c) {
    p[c] = (float)(o[3 * r + c] + d[3 * r + c] * (double)dist);
    pts[3 * i + c] = (double)p[c];
  }
  outside[i] = (uint8_t)((lo[0] > (double)p[0]) | (lo[1] > (double)p[1]) | (lo[2] > (double)p[2]) | (hi[0] < (double)p[0]) |
                         (hi[1] < (double)p[1]) | (hi[2] < (double)p[2]));
}

// render_utils_kernel.cu:326-341: every intermediate is a FLOAT variable; only the loads are double
__global__ void k_bg(const double *o, const double *d, const double *t_max, float keep, int n_samples, int64_t total, double *pts) {
  const int64_t i = gid();
  if (i >= total) return;
  const int64_t r = i / n_samples;
  const int s = (int)(i - r * n_samples);
  const float t_in = (float)t_max[r];
  const float frac = ((float)s) / (float)n_samples;
  const float t_out = (float)((double)t_in - 1. + 1. / (1. - (double)frac));
  const float x = (float)(o[3 * r] + d[3 * r] * (double)t_out);
  const float y = (float)(o[3 * r + 1] + d[3 * r + 1] * (double)t_out);
  const float z = (float)(o[3 * r + 2] + d[3 * r + 2] * (double)t_out);
  const float tn = sqrtf(x * x + y * y + z * z);
  const float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
  const float Ro = tn / m;
  const float q = (float)((double)(Ro * Ro / (tn * tn)) * (1. - (double)keep) + (double)(Ro / tn * keep));
  pts[3 * i] = (double)(x * q);
  pts[3 * i + 1] = (double)(y * q);
  pts[3 * i + 2] = (double)(z * q);
}

// render_utils_kernel.cu:385-390: round() of the double product, converted to int by the hardware (saturating, NaN -> 0);
// written without the out-of-range conversion, which C++ leaves undefined
__global__ void k_mask(const uint8_t *world, const double *xyz, const double *scale, const double *shift, int64_t ni, int64_t nj, int64_t nk,
                       int64_t n, uint8_t *out) {
  const int64_t p = gid();
  if (p >= n) return;
  double f[3];
  for (int c = 0; c < 3; ++c) {
    f[c] = round(xyz[3 * p + c] * scale[c] + shift[c]);
    f[c] = (f[c] != f[c]) ? 0.0 : f[c];          // the conversion maps NaN to 0: a NaN coordinate indexes plane 0 of its axis
  }
  const bool in = f[0] >= 0.0 && f[0] < (double)ni && f[1] >= 0.0 && f[1] < (double)nj && f[2] >= 0.0 && f[2] < (double)nk;
  out[p] = in ? world[((int64_t)f[0] * nj + (int64_t)f[1]) * nk + (int64_t)f[2]] : (uint8_t)0;
}

// ---- density -> alpha ----------------------------------------------------------------------------------------------------
// render_utils_kernel.cu:439-441, 454-456 (shift and the uniform interval are float ARGUMENTS)
__global__ void k_alpha(const double *density, float shift, float interval, const double *interval_arr, int64_t n, double *exp_d, double *alpha) {
  const int64_t i = gid();
  if (i >= n) return;
  const double e = exp(density[i] + (double)shift);
  exp_d[i] = e;
  alpha[i] = 1 - pow(1 + e, interval_arr ? -interval_arr[i] : (double)(-interval));
}

// render_utils_kernel.cu:516, 528: ((min(e, 1e10) * pow(1 + e, -interval - 1)) * interval) * grad_back; `-interval - 1` is formed in
// float when the interval is the scalar argument
__global__ void k_alpha_bwd(const double *exp_d, const double *grad_back, float interval, const double *interval_arr, int64_t n, double *grad) {
  const int64_t i = gid();
  if (i >= n) return;
  const double e = exp_d[i];
  const double iv = interval_arr ? interval_arr[i] : (double)interval;
  const double ex = interval_arr ? -interval_arr[i] - 1 : (double)(-interval - 1);
  grad[i] = fmin(e, 1e10) * pow(1 + e, ex) * iv * grad_back[i];
}

// ---- alpha -> weights ----------------------------------------------------------------------------------------------------
__global__ void k_fill_f64(double *a, double va, double *b, double vb, int64_t n) {
  const int64_t i = gid();
  if (i >= n) return;
  a[i] = va;
  if (b) b[i] = vb;
}

// segment of every ray in the sorted id list: ray r owns [first index with id >= r, first index with id > r) -- empty rays get 0, 0
// like the reference's zero-initialised arrays (render_utils_kernel.cu:607-617, 628-635)
__global__ void k_segments(const int64_t *ray_id, int64_t n, int64_t n_rays, int64_t *i_start, int64_t *i_end) {
  const int64_t r = gid();
  if (r >= n_rays) return;
  int64_t a = 0, b = n;
  while (a < b) {
    const int64_t m = (a + b) >> 1;
    if (ray_id[m] < r) a = m + 1; else b = m;
  }
  const int64_t first = a;
  b = n;
  while (a < b) {
    const int64_t m = (a + b) >> 1;
    if (ray_id[m] <= r) a = m + 1; else b = m;
  }
  const bool any = a > first;
  i_start[r] = any ? first : 0;
  i_end[r] = any ? a : 0;
}

// render_utils_kernel.cu:590-602: T_cum is a FLOAT whatever the tensors are; the product (1. - alpha) is double
__global__ void k_weights(const double *alpha, int64_t n_rays, double *weight, double *T, double *last, const int64_t *i_start, int64_t *i_end) {
  const int64_t r = gid();
  if (r >= n_rays) return;
  const int64_t lo = i_start[r], hi = i_end[r];
  float t_cum = 1.f;
  int64_t i = lo;
  for (; i < hi; ++i) {
    T[i] = (double)t_cum;
    weight[i] = (double)t_cum * alpha[i];
    t_cum = (float)((double)t_cum * (1. - alpha[i]));
    if ((double)t_cum < 1e-3) {
      ++i;
      break;
    }
  }
  i_end[r] = i;
  last[r] = (double)t_cum;
}

// render_utils_kernel.cu:671-675: back_cum is a FLOAT
__global__ void k_weights_bwd(const double *alpha, const double *weight, const double *T, const double *last, const int64_t *i_start,
                              const int64_t *i_end, int64_t n_rays, const double *g_w, const double *g_last, double *grad) {
  const int64_t r = gid();
  if (r >= n_rays) return;
  float back = (float)(g_last[r] * last[r]);
  for (int64_t i = i_end[r] - 1; i >= i_start[r]; --i) {
    grad[i] = g_w[i] * T[i] - (double)back / (1 - alpha[i] + 1e-10);
    back = (float)((double)back + g_w[i] * weight[i]);
  }
}

// ---- total variation -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ double unit_clamp(double v) { return fmin(fmax(v, (double)-1.f), (double)1.f); }

// total_variation_kernel.cu:24-37: the accumulator is a FLOAT (each term, a double, is added and rounded), the x-axis term is
// weighted by wz like the other two z-terms (the reference's quirk), an edge contributes the double 0
__global__ void k_tv(const double *param, double *grad, float wy, float wz, bool dense, int64_t ni, int64_t nj, int64_t nk, int64_t N) {
  const int64_t e = gid();
  if (e >= N) return;
  if (!dense && grad[e] == 0.0) return;
  const int64_t k = e % nk, j = e / nk % nj, i = e / nk / nj % ni;
  const double c = param[e];
  float acc = 0.f;
  auto add = [&](bool edge, float w, int64_t off) { acc = (float)((double)acc + (edge ? 0.0 : (double)w * unit_clamp(c - param[e + off]))); };
  add(k == 0, wz, -1);
  add(k == nk - 1, wz, 1);
  add(j == 0, wy, -nk);
  add(j == nj - 1, wy, nk);
  add(i == 0, wz, -nk * nj);
  add(i == ni - 1, wz, nk * nj);
  grad[e] += (double)acc;
}

// ub360_utils_kernel.cu:22-32: the running distance is a FLOAT
__global__ void k_cumdist(const double *dist, float thres, int64_t n_rays, int64_t n_pts, uint8_t *mask) {
  const int64_t r = gid();
  if (r >= n_rays) return;
  float run = 0.f;
  for (int64_t i = r * n_pts; i < (r + 1) * n_pts; ++i) {
    run = (float)((double)run + dist[i]);
    const bool over = run > thres;
    run *= (float)(!over);
    mask[i] = (uint8_t)over;
  }
}

// adam_upd_kernel.cu:19-21, 36-38, 54-56: betas, eps and the step size are float arguments, (1 - beta) is formed in float
template <int MODE>
__global__ void k_adam(double *param, const double *grad, double *m1, double *m2, const double *perlr, int64_t N, float step_size, float beta1,
                       float beta2, float eps) {
  const int64_t e = gid();
  if (e >= N) return;
  const double g = grad[e];
  if (MODE == 1 && g == 0.0) return;
  const double m = (double)beta1 * m1[e] + (double)(1 - beta1) * g;
  const double v = (double)beta2 * m2[e] + (double)(1 - beta2) * g * g;
  m1[e] = m;
  m2[e] = v;
  if (MODE == 2) param[e] -= (double)step_size * perlr[e] * m / (sqrt(v) + (double)eps);
  else param[e] -= (double)step_size * m / (sqrt(v) + (double)eps);
}

}  // namespace

#define GO(kernel, n, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid_for(n)), dim3(kThreads), 0, S(stream), __VA_ARGS__)

extern "C" int ugrid_infer_t_minmax_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max, float near,
                                        float far, int64_t n_rays, double *t_min, double *t_max, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  GO(k_span, n_rays, s, rays_o, rays_d, xyz_min, xyz_max, near, far, n_rays, t_min, t_max);
  return launched();
}

extern "C" int ugrid_infer_n_samples_f64(const double *rays_d, const double *t_min, const double *t_max, float stepdist, int64_t n_rays,
                                         int64_t *n_samples, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  GO(k_steps, n_rays, s, rays_d, t_min, t_max, stepdist, n_rays, n_samples);
  return launched();
}

extern "C" int ugrid_infer_ray_start_dir_f64(const double *rays_o, const double *rays_d, const double *t_min, int64_t n_rays, double *rays_start,
                                             double *rays_dir, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  GO(k_start_dir, n_rays, s, rays_o, rays_d, t_min, n_rays, rays_start, rays_dir);
  return launched();
}

extern "C" int ugrid_sample_pts_on_rays_count_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                                  float near, float far, float stepdist, int64_t n_rays, double *t_min, double *t_max,
                                                  int64_t *n_steps, int64_t *n_steps_cumsum, int64_t *d_total, ugrid_stream_t s) {
  if (n_rays <= 0) return (int)hipMemsetAsync(d_total, 0, sizeof(int64_t), S(s));
  GO(k_span_steps, n_rays, s, rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist, n_rays, t_min, t_max, n_steps);
  hipLaunchKernelGGL(k_prefix, dim3(1), dim3(1024), 0, S(s), n_steps, n_rays, n_steps_cumsum, d_total);
  return launched();
}

extern "C" int ugrid_sample_pts_on_rays_fill_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                                 const double *t_min, const int64_t *n_steps_cumsum, float stepdist, int64_t n_rays,
                                                 int64_t total_len, double *rays_pts, uint8_t *mask_outbbox, int64_t *ray_id, int64_t *step_id,
                                                 ugrid_stream_t s) {
  if (total_len <= 0) return 0;
  GO(k_fill, total_len, s, rays_o, rays_d, xyz_min, xyz_max, t_min, n_steps_cumsum, stepdist, n_rays, total_len, rays_pts, mask_outbbox, ray_id,
     step_id);
  return launched();
}

extern "C" int ugrid_sample_ndc_pts_on_rays_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                                int64_t n_samples, int64_t n_rays, double *rays_pts, uint8_t *mask_outbbox, ugrid_stream_t s) {
  const int64_t total = n_samples * n_rays;
  if (total <= 0) return 0;
  GO(k_ndc, total, s, rays_o, rays_d, xyz_min, xyz_max, (int)n_samples, total, rays_pts, mask_outbbox);
  return launched();
}

extern "C" int ugrid_sample_bg_pts_on_rays_f64(const double *rays_o, const double *rays_d, const double *t_max, float bg_preserve,
                                               int64_t n_samples, int64_t n_rays, double *rays_pts, ugrid_stream_t s) {
  const int64_t total = n_samples * n_rays;
  if (total <= 0) return 0;
  GO(k_bg, total, s, rays_o, rays_d, t_max, bg_preserve, (int)n_samples, total, rays_pts);
  return launched();
}

extern "C" int ugrid_maskcache_lookup_f64(const uint8_t *world, const double *xyz, const double *xyz2ijk_scale, const double *xyz2ijk_shift,
                                          int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n_pts, uint8_t *out, ugrid_stream_t s) {
  if (n_pts <= 0) return 0;
  GO(k_mask, n_pts, s, world, xyz, xyz2ijk_scale, xyz2ijk_shift, sz_i, sz_j, sz_k, n_pts, out);
  return launched();
}

extern "C" int ugrid_raw2alpha_f64(const double *density, float shift, float interval, const double *interval_arr, int64_t n, double *exp_d,
                                   double *alpha, ugrid_stream_t s) {
  if (n <= 0) return 0;
  GO(k_alpha, n, s, density, shift, interval, interval_arr, n, exp_d, alpha);
  return launched();
}

extern "C" int ugrid_raw2alpha_backward_f64(const double *exp_d, const double *grad_back, float interval, const double *interval_arr, int64_t n,
                                            double *grad, ugrid_stream_t s) {
  if (n <= 0) return 0;
  GO(k_alpha_bwd, n, s, exp_d, grad_back, interval, interval_arr, n, grad);
  return launched();
}

extern "C" int ugrid_alpha2weight_f64(const double *alpha, const int64_t *ray_id, int64_t n, int64_t n_rays, double *weight, double *T,
                                      double *alphainv_last, int64_t *i_start, int64_t *i_end, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  if (n > 0) GO(k_fill_f64, n, s, weight, 0.0, T, 1.0, n);
  GO(k_segments, n_rays, s, ray_id, n, n_rays, i_start, i_end);
  GO(k_weights, n_rays, s, alpha, n_rays, weight, T, alphainv_last, i_start, i_end);
  return launched();
}

extern "C" int ugrid_alpha2weight_backward_f64(const double *alpha, const double *weight, const double *T, const double *alphainv_last,
                                               const int64_t *i_start, const int64_t *i_end, int64_t n, int64_t n_rays,
                                               const double *grad_weights, const double *grad_last, double *grad, ugrid_stream_t s) {
  if (n > 0) GO(k_fill_f64, n, s, grad, 0.0, (double *)nullptr, 0.0, n);
  if (n_rays <= 0) return launched();
  GO(k_weights_bwd, n_rays, s, alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last, grad);
  return launched();
}

extern "C" int ugrid_total_variation_add_grad_f64(const double *param, double *grad, float wx, float wy, float wz, int dense_mode, int64_t sz_i,
                                                  int64_t sz_j, int64_t sz_k, int64_t N, ugrid_stream_t s) {
  (void)wx;                                          // accepted and unused, like the reference (its x-axis term uses wz)
  if (N <= 0) return 0;
  GO(k_tv, N, s, param, grad, wy / 6, wz / 6, dense_mode != 0, sz_i, sz_j, sz_k, N);      // total_variation_kernel.cu:46-48
  return launched();
}

extern "C" int ugrid_cumdist_thres_f64(const double *dist, float thres, int64_t n_rays, int64_t n_pts, uint8_t *mask, ugrid_stream_t s) {
  if (n_rays <= 0 || n_pts <= 0) return 0;
  GO(k_cumdist, n_rays, s, dist, thres, n_rays, n_pts, mask);
  return launched();
}

extern "C" int ugrid_adam_upd_f64(double *param, const double *grad, double *exp_avg, double *exp_avg_sq, const double *perlr, int64_t N,
                                  int step, float beta1, float beta2, float lr, float eps, int mode, ugrid_stream_t s) {
  if (mode < 0 || mode > 2 || (mode == 2 && !perlr)) return (int)hipErrorInvalidValue;
  if (N <= 0) return 0;
  // adam_upd_kernel.cu:69: a float expression on the host (powf / sqrtf of the float arguments)
  const float step_size = lr * sqrtf(1 - powf(beta2, (float)step)) / (1 - powf(beta1, (float)step));
  if (mode == 0) GO(k_adam<0>, N, s, param, grad, exp_avg, exp_avg_sq, perlr, N, step_size, beta1, beta2, eps);
  else if (mode == 1) GO(k_adam<1>, N, s, param, grad, exp_avg, exp_avg_sq, perlr, N, step_size, beta1, beta2, eps);
  else GO(k_adam<2>, N, s, param, grad, exp_avg, exp_avg_sq, perlr, N, step_size, beta1, beta2, eps);
  return launched();
}
