/*
 * oracle/ref_ops_f64.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * Scalar, single-threaded CPU restatement of the DOUBLE instantiation of the reference's four native extension modules
 * (the .cu files under FourierGrid/cuda of sjtuytc/UnboundedNeRFPytorch dispatch AT_DISPATCH_FLOATING_TYPES: float, restated
 * in ref_ops.c, and double, restated here).  Same function list as ref_ops.c, names orc64_*, arrays double.
 *
 * With scalar_t = double the reference does NOT compute "everything in double": its kernels declare many intermediates as
 * `float` whatever the tensor type is (the slab distances and ray parameters, rnorm, the sample positions, T_cum and back_cum,
 * the TV accumulator, the running distance) and take their scalar arguments as `const float`.  C's usual arithmetic
 * conversions applied to the CUDA source give: `float * double` is a double; `float x = <double expression>` and
 * `float x += <double>` round to float.  Every function below writes those conversions out and cites the lines that fix them.
 *
 * PARITY STATUS: PINNED on the reference's own kernels called with double tensors: tests/golden/native_ops_f64.npz holds the
 * outputs of all 18 exported functions of oracle/_ref (the reference's .cu compiled for gfx950 by oracle/build_ref.py) on an
 * MI355X (tests/golden/gen_native_golden_f64.py; the -ffp-contract=off and the default-contraction builds agree bit for bit),
 * and tests/test_oracle_golden.py::test_c_oracle_f64_pinned_on_reference_kernels checks this file against them: bit-exact
 * everywhere except the four raw2alpha functions, whose exp / pow come from glibc here and from the device libm there (a few
 * ulp).  The HIP twins (include/ugrid_hip_f64.h) are pinned on the same file on the GPU (tests/test_gpu_ref_native.py).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile), into oracle/_build/liboracle.so beside ref_ops.c.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* render_utils_kernel.cu:22-34: vx .. bz are float VARIABLES; (max - o) / v is a double division (double operands) */
static void span64(const double *o, const double *d, const double *lo, const double *hi, float near, float far, float *t0, float *t1) {
  float v[3], a[3], b[3];
  for (int c = 0; c < 3; ++c) {
    v[c] = (float)((d[c] == 0) ? 1e-6 : d[c]);
    a[c] = (float)((hi[c] - o[c]) / v[c]);
    b[c] = (float)((lo[c] - o[c]) / v[c]);
  }
  *t0 = fmaxf(fminf(fmaxf(fmaxf(fminf(a[0], b[0]), fminf(a[1], b[1])), fminf(a[2], b[2])), far), near);
  *t1 = fmaxf(fminf(fminf(fminf(fmaxf(a[0], b[0]), fmaxf(a[1], b[1])), fmaxf(a[2], b[2])), far), near);
}

/* :48-51, :70-73  `const float rnorm = sqrt(<double sum>)` */
static float rnorm64(const double *d) { return (float)sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); }

/* render_utils_kernel.cu:12-35 */
ORC_API void orc64_infer_t_minmax(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                  float near, float far, int64_t n_rays, double *t_min, double *t_max) {
  for (int64_t r = 0; r < n_rays; ++r) {
    float t0, t1;
    span64(rays_o + 3 * r, rays_d + 3 * r, xyz_min, xyz_max, near, far, &t0, &t1);
    t_min[r] = t0;
    t_max[r] = t1;
  }
}

/* :38-55  max(ceil((t_max - t_min) * rnorm / stepdist), 1.): double product, float rnorm and stepdist promoted */
ORC_API void orc64_infer_n_samples(const double *rays_d, const double *t_min, const double *t_max, float stepdist, int64_t n_rays,
                                   int64_t *n_samples) {
  for (int64_t r = 0; r < n_rays; ++r) {
    const double c = ceil((t_max[r] - t_min[r]) * rnorm64(rays_d + 3 * r) / stepdist);
    n_samples[r] = (int64_t)(c > 1. ? c : 1.);
  }
}

/* :58-79 */
ORC_API void orc64_infer_ray_start_dir(const double *rays_o, const double *rays_d, const double *t_min, int64_t n_rays,
                                       double *rays_start, double *rays_dir) {
  for (int64_t r = 0; r < n_rays; ++r) {
    const float rn = rnorm64(rays_d + 3 * r);
    for (int c = 0; c < 3; ++c) {
      rays_start[3 * r + c] = rays_o[3 * r + c] + rays_d[3 * r + c] * t_min[r];
      rays_dir[3 * r + c] = rays_d[3 * r + c] / rn;
    }
  }
}

/* :193-213 (host side of sample_pts_on_rays up to the total) */
ORC_API int64_t orc64_sample_pts_on_rays_count(const double *rays_o, const double *rays_d, const double *xyz_min,
                                               const double *xyz_max, float near, float far, float stepdist, int64_t n_rays,
                                               double *t_min, double *t_max, int64_t *n_steps) {
  orc64_infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far, n_rays, t_min, t_max);
  orc64_infer_n_samples(rays_d, t_min, t_max, stepdist, n_rays, n_steps);
  int64_t total = 0;
  for (int64_t r = 0; r < n_rays; ++r) total += n_steps[r];
  return total;
}

/* :165-191 -- dist, px, py, pz are float; rays_start / rays_dir are the double arrays of infer_ray_start_dir */
ORC_API void orc64_sample_pts_on_rays_fill(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                           const double *t_min, const int64_t *n_steps, float stepdist, int64_t n_rays,
                                           int64_t total_len, double *rays_pts, uint8_t *mask_outbbox, int64_t *ray_id,
                                           int64_t *step_id) {
  int64_t idx = 0;
  for (int64_t r = 0; r < n_rays; ++r) {
    const float rn = rnorm64(rays_d + 3 * r);
    for (int64_t s = 0; s < n_steps[r] && idx < total_len; ++s, ++idx) {
      const float dist = stepdist * (int)s;
      float p[3];
      for (int c = 0; c < 3; ++c) {
        const double start = rays_o[3 * r + c] + rays_d[3 * r + c] * t_min[r];
        const double dir = rays_d[3 * r + c] / rn;
        p[c] = (float)(start + dir * dist);
        rays_pts[3 * idx + c] = p[c];
      }
      mask_outbbox[idx] = (uint8_t)((xyz_min[0] > p[0]) | (xyz_min[1] > p[1]) | (xyz_min[2] > p[2]) | (xyz_max[0] < p[0]) |
                                    (xyz_max[1] < p[1]) | (xyz_max[2] < p[2]));
      ray_id[idx] = r;
      step_id[idx] = s;
    }
  }
}

/* :245-270 */
ORC_API void orc64_sample_ndc_pts_on_rays(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                          int64_t n_samples, int64_t n_rays, double *rays_pts, uint8_t *mask_outbbox) {
  for (int64_t r = 0; r < n_rays; ++r)
    for (int64_t s = 0; s < n_samples; ++s) {
      const int64_t idx = r * n_samples + s;
      const float dist = ((float)(int)s) / (int)(n_samples - 1);
      float p[3];
      for (int c = 0; c < 3; ++c) {
        p[c] = (float)(rays_o[3 * r + c] + rays_d[3 * r + c] * dist);
        rays_pts[3 * idx + c] = p[c];
      }
      mask_outbbox[idx] = (uint8_t)((xyz_min[0] > p[0]) | (xyz_min[1] > p[1]) | (xyz_min[2] > p[2]) | (xyz_max[0] < p[0]) |
                                    (xyz_max[1] < p[1]) | (xyz_max[2] < p[2]));
    }
}

/* :301-345 -- every intermediate is a float variable; norm3 is instantiated for float there (sqrtf) */
ORC_API void orc64_sample_bg_pts_on_rays(const double *rays_o, const double *rays_d, const double *t_max, float bg_preserve,
                                         int64_t n_samples, int64_t n_rays, double *rays_pts) {
  for (int64_t r = 0; r < n_rays; ++r)
    for (int64_t s = 0; s < n_samples; ++s) {
      const int64_t idx = r * n_samples + s;
      const float t_inner = (float)t_max[r];
      const float ori_t_outer = (float)(t_inner - 1. + 1. / (1. - ((float)(int)s) / (int)n_samples));
      const float x = (float)(rays_o[3 * r] + rays_d[3 * r] * ori_t_outer);
      const float y = (float)(rays_o[3 * r + 1] + rays_d[3 * r + 1] * ori_t_outer);
      const float z = (float)(rays_o[3 * r + 2] + rays_d[3 * r + 2] * ori_t_outer);
      const float t_outer = sqrtf(x * x + y * y + z * z);
      const float m = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
      const float R_outer = t_outer / m;
      const float o2i_p = (float)(R_outer * R_outer / (t_outer * t_outer) * (1. - bg_preserve) + R_outer / t_outer * bg_preserve);
      rays_pts[3 * idx] = x * o2i_p;
      rays_pts[3 * idx + 1] = y * o2i_p;
      rays_pts[3 * idx + 2] = z * o2i_p;
    }
}

/* :367-392 -- `const int i = round(<double>)`: the hardware conversion saturates and maps NaN to 0 (pinned by the golden) */
ORC_API void orc64_maskcache_lookup(const uint8_t *world, const double *xyz, const double *scale, const double *shift,
                                    int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n_pts, uint8_t *out) {
  for (int64_t p = 0; p < n_pts; ++p) {
    double f[3];
    for (int c = 0; c < 3; ++c) {
      f[c] = round(xyz[3 * p + c] * scale[c] + shift[c]);
      if (f[c] != f[c]) f[c] = 0.0;
    }
    const int in = f[0] >= 0.0 && f[0] < (double)sz_i && f[1] >= 0.0 && f[1] < (double)sz_j && f[2] >= 0.0 && f[2] < (double)sz_k;
    out[p] = in ? world[((int64_t)f[0] * sz_j + (int64_t)f[1]) * sz_k + (int64_t)f[2]] : 0;
  }
}

/* :431-457 (shift and the uniform interval are float arguments) */
ORC_API void orc64_raw2alpha(const double *density, float shift, float interval, const double *interval_arr, int64_t n,
                             double *exp_d, double *alpha) {
  for (int64_t i = 0; i < n; ++i) {
    const double e = exp(density[i] + shift);
    exp_d[i] = e;
    alpha[i] = 1 - pow(1 + e, interval_arr ? -interval_arr[i] : (double)(-interval));
  }
}

/* :507-530 -- ((min(e, 1e10) * pow(1 + e, -interval - 1)) * interval) * grad_back; `-interval - 1` is a float expression when the
 * interval is the scalar argument */
ORC_API void orc64_raw2alpha_backward(const double *exp_d, const double *grad_back, float interval, const double *interval_arr,
                                      int64_t n, double *grad) {
  for (int64_t i = 0; i < n; ++i) {
    const double e = exp_d[i];
    const double iv = interval_arr ? interval_arr[i] : (double)interval;
    const double ex = interval_arr ? -interval_arr[i] - 1 : (double)(-interval - 1);
    grad[i] = fmin(e, 1e10) * pow(1 + e, ex) * iv * grad_back[i];
  }
}

/* :577-651 -- T_cum is a float (:588); host glue: zeros / ones outputs, segment ends from the sorted ray_id */
ORC_API void orc64_alpha2weight(const double *alpha, const int64_t *ray_id, int64_t n, int64_t n_rays, double *weight, double *T,
                                double *alphainv_last, int64_t *i_start, int64_t *i_end) {
  for (int64_t i = 0; i < n; ++i) { weight[i] = 0.; T[i] = 1.; }
  for (int64_t r = 0; r < n_rays; ++r) { alphainv_last[r] = 1.; i_start[r] = 0; i_end[r] = 0; }
  if (n == 0) return;
  for (int64_t i = 1; i < n; ++i)
    if (ray_id[i] != ray_id[i - 1]) { i_start[ray_id[i]] = i; i_end[ray_id[i - 1]] = i; }
  i_end[ray_id[n - 1]] = n;
  for (int64_t r = 0; r < n_rays; ++r) {
    float T_cum = 1.f;
    int64_t i = i_start[r];
    for (; i < i_end[r]; ++i) {
      T[i] = T_cum;
      weight[i] = T_cum * alpha[i];
      T_cum = (float)(T_cum * (1. - alpha[i]));
      if (T_cum < 1e-3) { ++i; break; }
    }
    i_end[r] = i;
    alphainv_last[r] = T_cum;
  }
}

/* :654-676 -- back_cum is a float */
ORC_API void orc64_alpha2weight_backward(const double *alpha, const double *weight, const double *T, const double *alphainv_last,
                                         const int64_t *i_start, const int64_t *i_end, int64_t n, int64_t n_rays,
                                         const double *grad_weights, const double *grad_last, double *grad) {
  for (int64_t i = 0; i < n; ++i) grad[i] = 0.;
  for (int64_t r = 0; r < n_rays; ++r) {
    float back_cum = (float)(grad_last[r] * alphainv_last[r]);
    for (int64_t i = i_end[r] - 1; i >= i_start[r]; --i) {
      grad[i] = grad_weights[i] * T[i] - back_cum / (1 - alpha[i] + 1e-10);
      back_cum = (float)(back_cum + grad_weights[i] * weight[i]);
    }
  }
}

/* total_variation_kernel.cu:14-67 -- grad_to_add is a float: every term (a double: float weight times the clamped double
 * difference, or the int 0 at an edge) is added and rounded; the x-axis terms use wz (the reference's quirk) */
static double clamp1(double v) { return fmin(fmax(v, (double)-1.f), (double)1.f); }

ORC_API void orc64_total_variation_add_grad(const double *param, double *grad, float wx, float wy, float wz, int dense_mode,
                                            int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t N) {
  wx /= 6; wy /= 6; wz /= 6; (void)wx;
  for (int64_t e = 0; e < N; ++e) {
    if (!dense_mode && grad[e] == 0) continue;
    const int64_t k = e % sz_k, j = e / sz_k % sz_j, i = e / sz_k / sz_j % sz_i;
    float g = 0;
    g = (float)(g + (k == 0 ? 0 : wz * clamp1(param[e] - param[e - 1])));
    g = (float)(g + (k == sz_k - 1 ? 0 : wz * clamp1(param[e] - param[e + 1])));
    g = (float)(g + (j == 0 ? 0 : wy * clamp1(param[e] - param[e - sz_k])));
    g = (float)(g + (j == sz_j - 1 ? 0 : wy * clamp1(param[e] - param[e + sz_k])));
    g = (float)(g + (i == 0 ? 0 : wz * clamp1(param[e] - param[e - sz_k * sz_j])));
    g = (float)(g + (i == sz_i - 1 ? 0 : wz * clamp1(param[e] - param[e + sz_k * sz_j])));
    grad[e] += g;
  }
}

/* ub360_utils_kernel.cu:13-33 -- cum_dist is a float */
ORC_API void orc64_cumdist_thres(const double *dist, float thres, int64_t n_rays, int64_t n_pts, uint8_t *mask) {
  for (int64_t r = 0; r < n_rays; ++r) {
    float cum = 0;
    for (int64_t i = r * n_pts; i < (r + 1) * n_pts; ++i) {
      cum = (float)(cum + dist[i]);
      const int over = cum > thres;
      cum *= (float)(!over);
      mask[i] = (uint8_t)over;
    }
  }
}

/* adam_upd_kernel.cu:9-58, 60-132: the step size is a float expression on the host, betas / eps float arguments; (1 - beta) float */
ORC_API void orc64_adam_upd(double *param, const double *grad, double *exp_avg, double *exp_avg_sq, const double *perlr, int64_t N,
                            int step, float beta1, float beta2, float lr, float eps, int mode) {
  const float step_size = lr * sqrtf(1 - powf(beta2, (float)step)) / (1 - powf(beta1, (float)step));
  for (int64_t i = 0; i < N; ++i) {
    if (mode == 1 && grad[i] == 0) continue;
    exp_avg[i] = beta1 * exp_avg[i] + (1 - beta1) * grad[i];
    exp_avg_sq[i] = beta2 * exp_avg_sq[i] + (1 - beta2) * grad[i] * grad[i];
    if (mode == 2) param[i] -= step_size * perlr[i] * exp_avg[i] / (sqrt(exp_avg_sq[i]) + eps);
    else param[i] -= step_size * exp_avg[i] / (sqrt(exp_avg_sq[i]) + eps);
  }
}
