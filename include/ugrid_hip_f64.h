/* fp64 twins of the drop-in ops of ugrid_hip.h -- libugrid_hip_f64.so (csrc/ugrid_ops_f64.hip).
 *
 * The reference's four extension modules dispatch on the tensors' type (AT_DISPATCH_FLOATING_TYPES:
 * render_utils_kernel.cu:92-708, total_variation_kernel.cu:50,59, ub360_utils_kernel.cu:40, adam_upd_kernel.cu:74,98,123).
 * Each entry point below is the double instantiation of the like-named one in ugrid_hip.h: same arguments, array pointers
 * double instead of float, scalar arguments float as in the reference's signatures (`const float near`, ...), same output
 * initialisation and error behaviour.  The arithmetic is the reference kernels' with scalar_t = double -- including the
 * intermediates they keep in float whatever the tensors are -- pinned bit for bit on those kernels compiled for gfx950
 * (tests/test_gpu_ref_native.py).  Not tuned: no caller on the rendering / training path passes doubles.
 * The scan of ugrid_sample_pts_on_rays_count_f64 needs no workspace. */
#ifndef UGRID_HIP_F64_H
#define UGRID_HIP_F64_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef UGRID_HIP_H
typedef void *ugrid_stream_t; /* hipStream_t */
#endif

int ugrid_infer_t_minmax_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max, float near,
                             float far, int64_t n_rays, double *t_min, double *t_max, ugrid_stream_t stream);
int ugrid_infer_n_samples_f64(const double *rays_d, const double *t_min, const double *t_max, float stepdist, int64_t n_rays,
                              int64_t *n_samples, ugrid_stream_t stream);
int ugrid_infer_ray_start_dir_f64(const double *rays_o, const double *rays_d, const double *t_min, int64_t n_rays,
                                  double *rays_start, double *rays_dir, ugrid_stream_t stream);
int ugrid_sample_pts_on_rays_count_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                       float near, float far, float stepdist, int64_t n_rays, double *t_min, double *t_max,
                                       int64_t *n_steps, int64_t *n_steps_cumsum, int64_t *d_total, ugrid_stream_t stream);
int ugrid_sample_pts_on_rays_fill_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                      const double *t_min, const int64_t *n_steps_cumsum, float stepdist, int64_t n_rays,
                                      int64_t total_len, double *rays_pts, uint8_t *mask_outbbox, int64_t *ray_id,
                                      int64_t *step_id, ugrid_stream_t stream);
int ugrid_sample_ndc_pts_on_rays_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                     int64_t n_samples, int64_t n_rays, double *rays_pts, uint8_t *mask_outbbox,
                                     ugrid_stream_t stream);
int ugrid_sample_bg_pts_on_rays_f64(const double *rays_o, const double *rays_d, const double *t_max, float bg_preserve,
                                    int64_t n_samples, int64_t n_rays, double *rays_pts, ugrid_stream_t stream);
int ugrid_maskcache_lookup_f64(const uint8_t *world, const double *xyz, const double *xyz2ijk_scale, const double *xyz2ijk_shift,
                               int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n_pts, uint8_t *out, ugrid_stream_t stream);
int ugrid_raw2alpha_f64(const double *density, float shift, float interval, const double *interval_arr, int64_t n, double *exp_d,
                        double *alpha, ugrid_stream_t stream);
int ugrid_raw2alpha_backward_f64(const double *exp_d, const double *grad_back, float interval, const double *interval_arr,
                                 int64_t n, double *grad, ugrid_stream_t stream);
int ugrid_alpha2weight_f64(const double *alpha, const int64_t *ray_id, int64_t n, int64_t n_rays, double *weight, double *T,
                           double *alphainv_last, int64_t *i_start, int64_t *i_end, ugrid_stream_t stream);
int ugrid_alpha2weight_backward_f64(const double *alpha, const double *weight, const double *T, const double *alphainv_last,
                                    const int64_t *i_start, const int64_t *i_end, int64_t n, int64_t n_rays,
                                    const double *grad_weights, const double *grad_last, double *grad, ugrid_stream_t stream);
int ugrid_total_variation_add_grad_f64(const double *param, double *grad, float wx, float wy, float wz, int dense_mode,
                                       int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t N, ugrid_stream_t stream);
int ugrid_cumdist_thres_f64(const double *dist, float thres, int64_t n_rays, int64_t n_pts, uint8_t *mask,
                            ugrid_stream_t stream);
/* mode 0 adam_upd, 1 masked_adam_upd, 2 adam_upd_with_perlr */
int ugrid_adam_upd_f64(double *param, const double *grad, double *exp_avg, double *exp_avg_sq, const double *perlr, int64_t N,
                       int step, float beta1, float beta2, float lr, float eps, int mode, ugrid_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UGRID_HIP_F64_H */
