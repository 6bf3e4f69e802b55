/*
 * ugrid_hip.h -- C ABI of libugrid_hip.so: the MI355X (gfx950) hot path of
 * sjtuytc/UnboundedNeRFPytorch's FourierGrid/DVGO renderer.
 *
 * Boundary rules
 *   - extern "C", plain device pointers + sizes, no torch types.  fp32 data, int64 indices,
 *     bool masks as uint8_t (torch.bool layout).  All pointers are DEVICE pointers of the
 *     current HIP device unless the parameter name starts with h_ (host).
 *   - every function enqueues on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream), never synchronises unless documented, and returns a hipError_t as int (0 = ok).
 *   - outputs are caller-allocated; "init" notes say what the function itself writes.
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * /root/reference/FourierGrid/cuda/).  The reference binds these through pybind11
 * (render_utils.cpp:170-184, total_variation.cpp:23, ub360_utils.cpp:21, adam_upd.cpp:79-86);
 * INTEGRATION.md shows the binding a maintainer adds to call this library instead.
 */
#ifndef UGRID_HIP_H
#define UGRID_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ugrid_stream_t; /* hipStream_t */

/* ABI version of this header (bumped on any signature change). */
int ugrid_abi_version(void);
/* "gfx950" -- the only architecture the code objects are built for. */
const char *ugrid_target_arch(void);

/* ------------------------------------------------------------------ render_utils_cuda */

/* replaces infer_t_minmax (render_utils.cpp:170 -> render_utils_kernel.cu:12-35,82-104).
 * rays_o, rays_d [n_rays,3]; xyz_min, xyz_max [3]; t_min, t_max [n_rays] written. */
int ugrid_infer_t_minmax(const float *rays_o, const float *rays_d, const float *xyz_min,
                         const float *xyz_max, float near, float far, int64_t n_rays,
                         float *t_min, float *t_max, ugrid_stream_t stream);

/* replaces infer_n_samples (render_utils.cpp:171 -> render_utils_kernel.cu:38-55,106-121). */
int ugrid_infer_n_samples(const float *rays_d, const float *t_min, const float *t_max,
                          float stepdist, int64_t n_rays, int64_t *n_samples, ugrid_stream_t stream);

/* replaces infer_ray_start_dir (render_utils.cpp:172 -> render_utils_kernel.cu:58-79,123-139). */
int ugrid_infer_ray_start_dir(const float *rays_o, const float *rays_d, const float *t_min,
                              int64_t n_rays, float *rays_start, float *rays_dir, ugrid_stream_t stream);

/* replaces sample_pts_on_rays (render_utils.cpp:173 -> render_utils_kernel.cu:144-242), split in
 * the two halves the reference separates with its N_steps.sum().item() host sync:
 *   _count: writes t_min,t_max [n_rays], n_steps [n_rays], the inclusive prefix sum
 *           n_steps_cumsum [n_rays] and *d_total (device int64) = total sample count.
 *           scan_ws: device scratch of ugrid_scan_ws_bytes(n_rays) bytes.
 *   _fill:  given total_len (read back by the caller from d_total) writes rays_pts [total,3],
 *           mask_outbbox [total] (uint8 bool), ray_id, step_id [total] (int64). */
int64_t ugrid_scan_ws_bytes(int64_t n);
int ugrid_sample_pts_on_rays_count(const float *rays_o, const float *rays_d, const float *xyz_min,
                                   const float *xyz_max, float near, float far, float stepdist,
                                   int64_t n_rays, float *t_min, float *t_max, int64_t *n_steps,
                                   int64_t *n_steps_cumsum, int64_t *d_total, void *scan_ws,
                                   ugrid_stream_t stream);
int ugrid_sample_pts_on_rays_fill(const float *rays_o, const float *rays_d, const float *xyz_min,
                                  const float *xyz_max, const float *t_min,
                                  const int64_t *n_steps_cumsum, float stepdist, int64_t n_rays,
                                  int64_t total_len, float *rays_pts, uint8_t *mask_outbbox,
                                  int64_t *ray_id, int64_t *step_id, ugrid_stream_t stream);

/* replaces sample_ndc_pts_on_rays (render_utils.cpp:174 -> render_utils_kernel.cu:245-293). */
int ugrid_sample_ndc_pts_on_rays(const float *rays_o, const float *rays_d, const float *xyz_min,
                                 const float *xyz_max, int64_t n_samples, int64_t n_rays,
                                 float *rays_pts, uint8_t *mask_outbbox, ugrid_stream_t stream);

/* replaces sample_bg_pts_on_rays (render_utils.cpp:175 -> render_utils_kernel.cu:301-360). */
int ugrid_sample_bg_pts_on_rays(const float *rays_o, const float *rays_d, const float *t_max,
                                float bg_preserve, int64_t n_samples, int64_t n_rays,
                                float *rays_pts, ugrid_stream_t stream);

/* replaces maskcache_lookup (render_utils.cpp:176 -> render_utils_kernel.cu:367-424).
 * world [sz_i,sz_j,sz_k] uint8 bool; xyz [n_pts,3]; out [n_pts] fully written (0 when out of range). */
int ugrid_maskcache_lookup(const uint8_t *world, const float *xyz, const float *xyz2ijk_scale,
                           const float *xyz2ijk_shift, int64_t sz_i, int64_t sz_j, int64_t sz_k,
                           int64_t n_pts, uint8_t *out, ugrid_stream_t stream);

/* replaces raw2alpha / raw2alpha_nonuni (render_utils.cpp:177-178 -> render_utils_kernel.cu:431-504).
 * interval_arr == NULL: uniform `interval`; else per-point intervals [n]. */
int ugrid_raw2alpha(const float *density, float shift, float interval, const float *interval_arr,
                    int64_t n, float *exp_d, float *alpha, ugrid_stream_t stream);

/* replaces raw2alpha_backward / raw2alpha_nonuni_backward (render_utils.cpp:179-180 ->
 * render_utils_kernel.cu:507-574). */
int ugrid_raw2alpha_backward(const float *exp_d, const float *grad_back, float interval,
                             const float *interval_arr, int64_t n, float *grad, ugrid_stream_t stream);

/* replaces alpha2weight (render_utils.cpp:181 -> render_utils_kernel.cu:577-651).
 * alpha [n]; ray_id [n] int64 sorted non-decreasing, values in [0,n_rays).  Writes EVERY element of
 * weight,T [n] (0 / 1 past the early stop), alphainv_last [n_rays] (1 for empty rays) and
 * i_start,i_end [n_rays] (0 for empty rays; i_end truncated at the early stop).  No host sync. */
int ugrid_alpha2weight(const float *alpha, const int64_t *ray_id, int64_t n, int64_t n_rays,
                       float *weight, float *T, float *alphainv_last, int64_t *i_start,
                       int64_t *i_end, ugrid_stream_t stream);

/* replaces alpha2weight_backward (render_utils.cpp:182 -> render_utils_kernel.cu:654-707).
 * grad [n] fully written (0 outside [i_start,i_end)). */
int ugrid_alpha2weight_backward(const float *alpha, const float *weight, const float *T,
                                const float *alphainv_last, const int64_t *i_start,
                                const int64_t *i_end, int64_t n, int64_t n_rays,
                                const float *grad_weights, const float *grad_last, float *grad,
                                ugrid_stream_t stream);

/* ------------------------------------------------------------------ total_variation_cuda */

/* replaces total_variation_add_grad (total_variation.cpp:23 -> total_variation_kernel.cu:14-67).
 * param, grad: N = (leading dims) * sz_i*sz_j*sz_k contiguous floats; grad updated in place.
 * wx is accepted and ignored exactly like the reference (its x-axis term uses wz). */
int ugrid_total_variation_add_grad(const float *param, float *grad, float wx, float wy, float wz,
                                   int dense_mode, int64_t sz_i, int64_t sz_j, int64_t sz_k,
                                   int64_t N, ugrid_stream_t stream);

/* ------------------------------------------------------------------ ub360_utils_cuda */

/* replaces cumdist_thres (ub360_utils.cpp:21 -> ub360_utils_kernel.cu:13-47). dist [n_rays,n_pts]. */
int ugrid_cumdist_thres(const float *dist, float thres, int64_t n_rays, int64_t n_pts,
                        uint8_t *mask, ugrid_stream_t stream);

/* segment_cumsum(w, s, ray_id) -> (w_prefix, w_total, ws_prefix, ws_total): the op the reference's DistortionLoss
 * calls (FourierGrid_model.py:689; also dcvgo.py:392) but its ub360_utils.cpp:21 never exports.  Exclusive fp32
 * running sums of w and w*s inside each segment of the sorted ray_id [n], in sample order, and per-ray totals
 * [n_rays] (0 for rays without samples).  seg_scratch: 2*n_rays int64 of device scratch. */
int ugrid_segment_cumsum(const float *w, const float *s, const int64_t *ray_id, int64_t n, int64_t n_rays,
                         float *w_prefix, float *w_total, float *ws_prefix, float *ws_total,
                         int64_t *seg_scratch, ugrid_stream_t stream);

/* ------------------------------------------------------------------ adam_upd_cuda */

/* replaces adam_upd / masked_adam_upd / adam_upd_with_perlr (adam_upd.cpp:79-86 ->
 * adam_upd_kernel.cu:9-132).  mode: 0 dense, 1 masked (skip grad==0), 2 per-voxel lr (perlr != NULL); 3 (no reference
 * counterpart) = masked + the gradient's nonzero elements are overwritten with 0 after use (it then serves as the next
 * backward's zero-initialised buffer; const-ness of `grad` is waived for this mode).
 * param, exp_avg, exp_avg_sq updated in place. */
/* NEW (no reference counterpart): dense total_variation_add_grad + (masked_)adam_upd in ONE pass over a grid parameter --
 * what run_train.py:281-288 does with two extension calls while global_step < tv_dense_before.  The TV term is added to
 * the gradient in registers (grad is not modified), the updated parameter is written to param_out (a second buffer of
 * the same shape: the stencil needs the neighbours' old values; the caller swaps the buffers), exp_avg / exp_avg_sq in
 * place.  Results are bit-identical to ugrid_total_variation_add_grad(dense) followed by ugrid_adam_upd(mode =
 * skip_zero_grad).  Returns hipErrorNotSupported (801) when the shape cannot take the vector path (sz_k % 4 != 0,
 * N >= 2^31, unaligned or aliasing buffers); the caller then uses the two separate entry points.
 * flags: bit 0 = skip_zero_grad (the masked_adam_upd rule on the TV-added gradient); bit 1 = rezero_grad: every nonzero
 * element of grad is overwritten with 0 after it has been read (only the touched cache lines are written), so the
 * caller can hand the buffer to the next backward as its zero-initialised gradient instead of filling a new one. */
int ugrid_tv_adam_dense(const float *param, float *param_out, const float *grad, float *exp_avg, float *exp_avg_sq,
                        float wx, float wy, float wz, int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t N, int step,
                        float beta1, float beta2, float lr, float eps, int flags, ugrid_stream_t stream);

int ugrid_adam_upd(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                   const float *perlr, int64_t N, int step, float beta1, float beta2, float lr,
                   float eps, int mode, ugrid_stream_t stream);

/* NEW (round 5, no reference counterpart): the reference's optimizer loops over the parameters with one extension call each
 * (masked_adam.py:43-75); this applies adam_upd (mode 0) or masked_adam_upd (mode 1) to n_items SMALL tensors -- the rgbnet's
 * six -- in one launch.  h_items is a HOST array; step / lr per tensor, betas / eps shared.  Element for element the arithmetic
 * of ugrid_adam_upd: bit-identical results. */
typedef struct {
  float *param;
  const float *grad;
  float *exp_avg, *exp_avg_sq;
  int64_t numel;
  int32_t step;
  float lr;
} ugrid_adam_item;
int ugrid_adam_upd_multi(const ugrid_adam_item *h_items, int32_t n_items, float beta1, float beta2, float eps, int32_t mode,
                         ugrid_stream_t stream);

/* ------------------------------------------------------------------ fused render path
 * New entry points (no native counterpart in the reference): they replace the torch-op chain of
 * FourierGridModel.forward (FourierGrid_model.py:554-672) and FourierGrid.forward
 * (FourierGrid_grid.py:60-78) for inference.  See DESIGN.md for layouts. */

/* Fourier / dense grid query: replaces FourierGrid.forward (FourierGrid_grid.py:60-78) and
 * DenseGrid.forward (grid.py:50-61) = F.grid_sample(bilinear, align_corners=True, zeros) + mean over
 * levels.  grid [P,C,X,Y,Z] canonical layout; xyz [n,3] world coords; out [n,C].  freq_num = F
 * (P must be 1+2F) or 0 for a single-level grid. */
int ugrid_grid_query(const float *grid, int P, int C, int X, int Y, int Z, const float *xyz,
                     const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                     float *out, ugrid_stream_t stream);

/* Gradient of ugrid_grid_query w.r.t. the grid (the autograd backward of FourierGrid.forward /
 * DenseGrid.forward, i.e. F.grid_sample's grid gradient followed by the mean over levels,
 * FourierGrid_grid.py:60-78): grad_grid [P,C,X,Y,Z] += scatter of grad_out [n,C] through the trilinear
 * weights (zero padding: taps outside the grid receive nothing).  grad_grid must be initialised by the caller
 * (zeros for a fresh gradient); hardware fp32 atomics, so sums agree with torch's to rounding.  Entries whose
 * incoming gradient is exactly 0 add nothing (MaskedAdam's skip_zero_grad test keeps working). */
int ugrid_grid_query_backward(const float *grad_out, int P, int C, int X, int Y, int Z, const float *xyz,
                              const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                              float *grad_grid, ugrid_stream_t stream);

/* Channel-last twins (round 2, training layout of multi-channel grids): `grid` / `grad_grid` are the SAME logical
 * [P,C,X,Y,Z] tensors stored as [P][X][Y][Z][C] (torch.channels_last_3d): the C channels of a voxel form one 4C-byte run,
 * so the C atomics of a corner in the scatter hit one cache line instead of C planes 4*X*Y*Z bytes apart.  Same
 * arithmetic and order per channel as the canonical entry points. */
int ugrid_grid_query_cl(const float *grid, int P, int C, int X, int Y, int Z, const float *xyz,
                        const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                        float *out, ugrid_stream_t stream);
int ugrid_grid_query_backward_cl(const float *grad_out, int P, int C, int X, int Y, int Z, const float *xyz,
                                 const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                 float *grad_grid, ugrid_stream_t stream);
/* total_variation_add_grad and the fused dense TV + Adam pass on channel-last storage [planes][sz_i][sz_j][sz_k][C]
 * (N = planes*sz_i*sz_j*sz_k*C); C % 4 == 0, N < 2^31 and 16-byte alignment required, else hipErrorNotSupported. */
int ugrid_total_variation_add_grad_cl(const float *param, float *grad, float wx, float wy, float wz, int dense_mode,
                                      int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t C, int64_t N,
                                      ugrid_stream_t stream);
int ugrid_tv_adam_dense_cl(const float *param, float *param_out, const float *grad, float *exp_avg, float *exp_avg_sq,
                           float wx, float wy, float wz, int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t C, int64_t N,
                           int step, float beta1, float beta2, float lr, float eps, int flags,
                           ugrid_stream_t stream);

/* Touched-line bitmap of a recycled gradient buffer (NEW, training; channel-last grids).  The lookup backward of a training
 * step touches a few per cent of the 256-byte lines of the grid-sized gradient (S3: ~5 % of 3.46 GB); the reference's masked
 * TV and masked Adam (total_variation_kernel.cu:14-67 masked, adam_upd_kernel.cu:26-41) find them by scanning the whole array
 * for non-zeros, twice per step.  Here the backward marks what it touches -- bit (e >> 6) & 31 of word e >> 11 for element e of
 * grad_grid, ugrid_touch_words(N) uint32 words, all zero initially -- and the passes below read the bitmap and only the
 * marked lines.  Contract: between the backward and the passes nobody writes a non-zero into an unmarked line (results are
 * then identical to the scanning kernels: within a marked line the per-element `grad != 0` rule still decides).
 *   ugrid_grid_query_backward_cl_touch       ugrid_grid_query_backward_cl + marking
 *   ugrid_total_variation_add_grad_cl_touch  masked mode (dense_mode = 0) on the marked lines
 *   ugrid_masked_adam_upd_touch              masked_adam_upd on the marked lines; grad comes back all zero, touch cleared
 *   ugrid_tv_adam_dense_cl_touch             the fused dense pass without reading unmarked gradient lines; with the rezero
 *                                            flag (flags & 2) touch is cleared as well */
int64_t ugrid_touch_words(int64_t N);
int ugrid_grid_query_backward_cl_touch(const float *grad_out, int P, int C, int X, int Y, int Z, const float *xyz,
                                       const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                       float *grad_grid, uint32_t *touch, ugrid_stream_t stream);
int ugrid_total_variation_add_grad_cl_touch(const float *param, float *grad, float wx, float wy, float wz,
                                            int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t C, int64_t N,
                                            const uint32_t *touch, ugrid_stream_t stream);
int ugrid_masked_adam_upd_touch(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t N, int step,
                                float beta1, float beta2, float lr, float eps, uint32_t *touch, ugrid_stream_t stream);
int ugrid_tv_adam_dense_cl_touch(const float *param, float *param_out, const float *grad, float *exp_avg, float *exp_avg_sq,
                                 float wx, float wy, float wz, int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t C, int64_t N,
                                 int step, float beta1, float beta2, float lr, float eps, int flags, uint32_t *touch,
                                 ugrid_stream_t stream);

/* The rgbnet of the training step (FourierGrid_model.py:233-241, :636: Linear(mlp_in,128)-ReLU-Linear(128,128)-ReLU-Linear(128,3)
 * on the M surviving samples) and its derivative as fp32-MFMA kernels (csrc/ugrid_train_mlp.hip), replacing the library GEMMs of
 * torch's nn.Linear forward / backward whose host overhead dominates at M ~ 1e5.  nn.Linear layouts: w0 [128, mlp_in], w1 [128,128],
 * w2 [3,128] (128 = `width`); feat [M, mlp_in] row-major.  forward: h1, h2 [M,128] (post-ReLU activations, kept for the backward),
 * logits [M,3].
 * backward: g_w*, g_b* (overwritten, not accumulated), g_feat [M, n_feat_grad] = the gradient of the first n_feat_grad input
 * columns (the k0 features; 0 / NULL = none); scratch: ugrid_rgbnet_train_scratch_floats(M) floats.  fp32-accurate products
 * (bf16x3 or fp32 MFMA: ugrid_tune "train_mlp"), fp32 accumulation; the weight gradients are sums of <= 256 slab partials added in a fixed order (deterministic).  width <= 128
 * (h1 / h2 are [M, width], w0 [width, mlp_in], w1 [width, width], w2 [3, width]), mlp_in <= 128. */
int64_t ugrid_rgbnet_train_scratch_floats(int64_t M);
int ugrid_rgbnet_train_forward(const float *feat, int64_t M, int32_t mlp_in, const float *w0, const float *b0, const float *w1,
                               const float *b1, const float *w2, const float *b2, int32_t width, float *h1, float *h2,
                               float *logits, ugrid_stream_t stream);
int ugrid_rgbnet_train_backward(const float *g_logits, const float *feat, const float *h1, const float *h2, int64_t M,
                                int32_t mlp_in, int32_t n_feat_grad, const float *w0, const float *w1, const float *w2,
                                int32_t width, float *g_feat, float *g_w0, float *g_b0, float *g_w1, float *g_b1, float *g_w2,
                                float *g_b2, float *scratch, ugrid_stream_t stream);

/* NEW (no reference counterpart; replaces the elementwise chain of FourierGrid_model.py:631-635 / dvgo.py:352-357): the rgbnet's
 * input rows [k0 | viewdir | sin(viewdir * viewfreq) | cos(viewdir * viewfreq)] of the m surviving samples.
 * k0 [m, n_k0] (NULL with n_k0 = 0: only the view embedding rows), viewdirs [n_rays, 3], viewfreq [pe] (2^k), ray_id [m] (NULL:
 * row i uses viewdirs[i]), out [m, n_k0 + 3 + 6 pe].  Column order is torch's cat([viewdirs, e.sin(), e.cos()]) with
 * e = (viewdirs[..., None] * viewfreq).flatten(-2).  sinf / cosf of the same fp32 product torch forms.
 * ray_rows: scratch [n_rays, 3 + 6 pe] or NULL.  With it (and ray_id, and m >= 2 n_rays) the embedding is formed once per ray and
 * gathered (two launches, the same values); without it every output element evaluates its own sine / cosine (one launch). */
int ugrid_rgbnet_features(const float *k0, int32_t n_k0, const float *viewdirs, int64_t n_rays, const float *viewfreq, int32_t pe,
                          const int64_t *ray_id, int64_t m, float *ray_rows, float *out, ugrid_stream_t stream);

/* NEW (no reference counterpart; training tail): sigmoid + per-ray compositing + the loss of run_train.py:254-279 in
 * one pass, and its derivative -- replaces FourierGrid_model.py:636-647 (sigmoid, weights * rgb, segment_coo, background)
 * and run_train.py:254-279 (mse_loss, entropy_last, nearclip, flatten_eff_distloss, rgbper): ~45 + ~90 launches of the
 * composed chain.  ray_id ascending (the model's compaction is ray-major); logits [n,3] are the rgbnet outputs before the
 * sigmoid; bg [n_rays,3] the rand_bkgd draw or NULL; target [n_rays,3]; s may be NULL: s_i = 1 - 1/(1 + t_i) then
 * (FourierGrid_model.py:649).
 * h_coef9 (9 HOST floats): weight_main, weight_entropy_last, weight_distortion, weight_rgbper, weight_nearclip (times the
 * data-parallel world size), near_thres, interval (= 1/n_max), n_rays (the rgbper denominator), weight_freq (the image-space
 * Fourier MSE of run_train.py:255 / FourierGrid_model.py:112-129: MSE between the real parts of the 3-point FFTs over the
 * colour axis).  A weight of 0 disables its term.  seg_scratch: 2*n_rays int64 (filled by the forward, read by the backward);
 * rgb_marched [n_rays,3]; ray_tot [n_rays,2] (per-ray sums of w and w*s); partial [n_rays,5]; out2 = {loss, mse} on the device.
 * backward: grad_loss = the incoming scalar gradient (device); writes g_logits [n,3], g_weights [n], g_alphainv_last
 * [n_rays], g_density [n] (the nearclip term's only effect). */
int ugrid_render_loss(const float *logits, const float *weights, const float *s, const float *t, const float *alphainv_last,
                      const float *bg, const float *target, const int64_t *ray_id, int64_t n, int64_t n_rays,
                      const float *h_coef9, int64_t *seg_scratch, float *rgb_marched, float *ray_tot, float *partial,
                      float *out2, ugrid_stream_t stream);
int ugrid_render_loss_backward(const float *logits, const float *weights, const float *s, const float *t,
                               const float *alphainv_last, const float *bg, const float *target, const int64_t *ray_id,
                               int64_t n, int64_t n_rays, const float *h_coef9, const int64_t *seg_scratch,
                               const float *rgb_marched, const float *ray_tot, const float *grad_loss, float *g_logits,
                               float *g_weights, float *g_alphainv_last, float *g_density, ugrid_stream_t stream);

/* Brick packing: canonical [P,C,X,Y,Z] -> cell-major 2x2x2 bricks, one contiguous record per
 * trilinear cell: [P*(X-1)(Y-1)(Z-1)][H halves][...] with
 *   C == 1 (density)        : H = 1, [8 entries]                          (32 B / cell)
 *   C  > 1 (rgbnet features): H = 2, [ceil(C/4) channel pairs][8 entries][2 channels]
 *                                                    (zero padded; 384 B / cell at C = 12)
 *   direct != 0 (no rgbnet, C == 3): H = 1, [8 corners][4 channels]       (128 B / cell)
 * For the first two the 8 entries of a channel are the coefficients of the cell's trilinear polynomial
 * sum c[4dx+2dy+dz] tx^dx ty^dy tz^dz in the fractional cell coordinates (computed in fp64 from the 8 corner
 * values, rounded once), so a lookup is 7 FMAs per channel; direct bricks hold the corner values.
 * ugrid_brick_bytes returns the size of the packed array. */
int64_t ugrid_brick_bytes(int P, int C, int X, int Y, int Z, int direct);
int ugrid_pack_bricks(const float *grid, int P, int C, int X, int Y, int Z, int direct, float *bricks,
                      ugrid_stream_t stream);

/* Parameters of the fused FourierGrid render (HOST struct, passed by pointer). */
typedef struct ugrid_render_params {
  int64_t n_rays;
  int32_t n_samples;              /* S = len(t) */
  int32_t freq_num;               /* F; P = 1+2F levels (1..5; 0 = single-level DenseGrid models: ugrid_render_march_dcvgo) */
  int32_t grid_x, grid_y, grid_z; /* density == k0 resolution */
  int32_t k0_channels;            /* C (12, or 3) */
  int32_t mlp_in;                 /* C + 3 + 6*viewbase_pe; 0 => no rgbnet (rgb = sigmoid(k0), C==3) */
  int32_t mlp_width;              /* 128 */
  int32_t viewbase_pe;
  int32_t norm_l2;                /* 0: inf-norm contraction, 1: l2 */
  float scene_center[3], scene_radius[3];
  float xyz_min[3], xyz_max[3];   /* contracted bounds (-1-bg .. 1+bg) as the fp32 buffers hold them */
  double bg_len;                  /* python float, e.g. 0.2 */
  float act_shift, interval, thres;
  int32_t mlp_mode;               /* rgbnet arithmetic of the shade kernel: UGRID_MLP_FP32 | _BF16X3 | _FP16X2
                                     (all fp32-accurate; FP16X2 only if ugrid_pack_mlp reported it usable) */
} ugrid_render_params;

#define UGRID_MLP_FP32 0    /* v_mfma_f32_32x32x2_f32: plain fp32 products */
#define UGRID_MLP_BF16X3 1  /* weights and activations split into 3 bf16 parts, 6 MFMA products (~2^-24) */
#define UGRID_MLP_FP16X2 2  /* power-of-two scaled operands split into 2 fp16 parts, 3 MFMA products (~2^-22) */
/* flag OR-ed into mlp_mode for ugrid_render_shade: residual colour of the reference's DirectVoxGO with rgbnet_direct = False
 * (dvgo.py:385-398): rgb = sigmoid(rgbnet([k0[3:], view embedding]) + k0[:3]).  The caller packs the first layer [128, C + emb]
 * with ZERO columns for the three diffuse channels (ugrid_pack_mlp sees a network of the usual shape); needs k0_channels >= 9. */
#define UGRID_MLP_RESIDUAL 0x100

/* Bytes of the survivor work list (worst case: every sample survives) for n_rays x n_samples. */
int64_t ugrid_render_ws_bytes(int64_t n_rays, int32_t n_samples);

/* Fused march: rays -> contracted samples -> P-level density bricks -> raw2alpha -> threshold ->
 * front-to-back scan with early stop -> threshold; writes alphainv_last [R], depth [R] and per-tile
 * (64 rays) survivor lists into ws.  t_table, s_table [S] device floats (sample distances t and
 * s = 1-1/(1+t), FourierGrid_model.py:524-532,649). */
int ugrid_render_march(const ugrid_render_params *h_params, const float *rays_o, const float *rays_d,
                       const float *t_table, const float *s_table, const float *density_bricks,
                       float *alphainv_last, float *depth, void *ws, ugrid_stream_t stream);

/* Fused march of the reference's DirectContractedVoxGO.forward (dcvgo.py:228-345; single-level grids, h_params->freq_num
 * = 0): the march above plus, per sample, the "skip oversampled points" rule -- a contracted sample is evaluated only
 * when the running sum of inter-sample distances has just exceeded dist_thres (ub360_utils_cuda.cumdist_thres,
 * ub360_utils_kernel.cu:24-31; dcvgo.py:283-289) -- and the mask cache (render_utils_cuda.maskcache_lookup,
 * render_utils_kernel.cu:374-392; grid.py:229-237: nearest voxel of a DEVICE bool grid [mask_x, mask_y, mask_z] at
 * round(p * xyz2ijk_scale + xyz2ijk_shift)).  Also writes wsum_mid [R], the weight sum of the surviving un-contracted
 * samples (dcvgo.py:354-358).  t_table / s_table: dcvgo's sample distances (boundary 2, dcvgo.py:243-250) and
 * s = 1 - 1/(1+t).  The survivor list in ws feeds ugrid_render_shade (freq_num = 0). */
typedef struct ugrid_dcvgo_params {
  const uint8_t *mask;              /* DEVICE bool [mask_x][mask_y][mask_z] (MaskGrid.mask, grid.py:221-228) */
  int32_t mask_x, mask_y, mask_z;
  float xyz2ijk_scale[3], xyz2ijk_shift[3];
  float dist_thres;                 /* (2 + 2 bg_len) / world_len * stepsize * 0.95 */
} ugrid_dcvgo_params;
int ugrid_render_march_dcvgo(const ugrid_render_params *h_params, const ugrid_dcvgo_params *h_dc, const float *rays_o,
                             const float *rays_d, const float *t_table, const float *s_table,
                             const float *density_bricks, float *alphainv_last, float *depth, float *wsum_mid,
                             void *ws, ugrid_stream_t stream);

/* Fused march of the reference's bounded DirectVoxGO.forward (dvgo.py:306-400; single-level grids, h_params->freq_num = 0,
 * h_params->xyz_min / xyz_max = the scene box): ray / box clipping (infer_t_minmax, render_utils_kernel.cu:16-41), per-ray
 * step counts (infer_n_samples :43-57) and point generation (sample_pts_on_rays :100-260) inside the march -- a lane loops
 * to its own ray's count, no count / cumsum / host read / fill --, mask_outbbox and the mask cache (maskcache_lookup :374-392)
 * before the lookup, then alpha, thresholds, compositing as above; depth = sum w * step_id (dvgo.py:419-423).
 * h_params->n_samples = an upper bound of the steps of a ray (sizes the survivor list in ws: the box diagonal / stepdist + 1;
 * rays are clamped to it).  The survivor list feeds ugrid_render_shade (freq_num = 0). */
typedef struct ugrid_dvgo_params {
  const uint8_t *mask;              /* DEVICE bool [mask_x][mask_y][mask_z] */
  int32_t mask_x, mask_y, mask_z;
  float xyz2ijk_scale[3], xyz2ijk_shift[3];
  float near_clip, far_clip;        /* render_kwargs['near'], 1e9 (dvgo.py:318) */
  float stepdist;                   /* stepsize * voxel_size */
} ugrid_dvgo_params;
int ugrid_render_march_dvgo(const ugrid_render_params *h_params, const ugrid_dvgo_params *h_dv, const float *rays_o,
                            const float *rays_d, const float *density_bricks, float *alphainv_last, float *depth,
                            void *ws, ugrid_stream_t stream);

/* Fused shade: survivors -> P-level k0 bricks -> [k0, viewdir emb] -> rgbnet (MFMA, h_params->mlp_mode) -> sigmoid
 * -> weighted per-ray sum in sample order; writes rgb_marched [R,3].  mlp_packed: ugrid_pack_mlp(). */
int ugrid_render_shade(const ugrid_render_params *h_params, const float *viewdirs,
                       const float *k0_bricks, const float *mlp_packed, void *ws,
                       float *rgb_marched, ugrid_stream_t stream);

/* rgbnet packing for the MFMA shade kernel: w0 [128, C+3+6pe], b0 [128], w1 [128,128], b1 [128],
 * w2 [3,128], b2 [3] (nn.Linear layout, FourierGrid_model.py:233-241) -> packed device array holding one
 * image per arithmetic mode.  k0_absmax: an upper bound on |k0 feature| (max |k0 grid value|: the feature is
 * a convex combination of grid values averaged over levels); it sizes the power-of-two activation scales of
 * the fp16x2 image (view directions are assumed unit length like the reference's, dvgo.py:517).  Pass <= 0
 * if unknown.  *best_mode (HOST int, may be NULL) receives the fastest usable mode: UGRID_MLP_FP16X2 when
 * the weights and the propagated activation bounds fit fp16's range after scaling and viewbase_pe <= 4 (the fp16x2
 * kernels keep a per-wave view-embedding table in LDS that does not fit beside the rgbnet image for wider embeddings),
 * else UGRID_MLP_BF16X3.  Synchronises the stream once (the weights are read back to derive the scales). */
int64_t ugrid_mlp_packed_bytes(int32_t k0_channels, int32_t viewbase_pe);
int ugrid_pack_mlp(const float *w0, const float *b0, const float *w1, const float *b1,
                   const float *w2, const float *b2, int32_t k0_channels, int32_t viewbase_pe,
                   int32_t width, float k0_absmax, float *packed, int32_t *best_mode,
                   ugrid_stream_t stream);

/* The host arithmetic behind ugrid_pack_mlp's answer, on HOST copies of w0 [128, C+3+6pe], b0 [128], w1 [128,128]:
 * scales4 = {sX1, sW1, sX2, sW2}, the power-of-two activation / weight scales of layers 1 and 2 of the fp16x2
 * image.  Returns 1 if the mode is usable (finite, non-degenerate ranges), else 0 (scales4 = 1).  No GPU needed. */
int ugrid_mlp_fp16x2_scales(const float *h_w0, const float *h_b0, const float *h_w1, int32_t k0_channels,
                            int32_t viewbase_pe, float k0_absmax, float *scales4);

/* ------------------------------------------------------------------ ray generation
 * replaces get_rays_of_a_view / get_rays (dvgo.py:493-521,554-559; twin FourierGrid_model.py:21-77), ndc=False: the ~15
 * torch launches of the elementwise chain as one kernel.  h_K9: HOST 3x3 intrinsics (row-major), c2w: DEVICE [3,4];
 * mode_center adds 0.5 to the pixel coordinates ('center'; 0 = 'lefttop').  pixel_index == NULL: all H*W pixels in image
 * order (n = H*W), else n flat indices j*W+i (a rank's shard of a frame).  Outputs [n,3] each. */
int ugrid_rays_of_a_view(int32_t H, int32_t W, const float *h_K9, const float *c2w, int inverse_y, int flip_x, int flip_y,
                         int mode_center, const int64_t *pixel_index, int64_t n, float *rays_o, float *rays_d,
                         float *viewdirs, ugrid_stream_t stream);

/* ------------------------------------------------------------------ fused training forward, stage 1 (new)
 * Replaces the head of FourierGridModel.forward in training mode (FourierGrid_model.py:554-598): sample_ray, the
 * density lookup on all R*S points, Raw2Alpha, the `alpha > fast_color_thres` mask and its boolean-index gathers.
 * ugrid_train_march: one wave per ray; every sample's point / density / alpha is formed with the operation order of the
 * composed path (torch elementwise chain of sample_ray, ugrid_grid_query, ugrid_raw2alpha) and the survivors of the
 * threshold are compacted into the ray's slot [r*S, r*S+count[r]) of three scratch arrays ([R*S,3] f32, [R*S] f32,
 * [R*S] i32).  host: offset_end = inclusive cumsum(count) (int64), M1 = offset_end[R-1].
 * ugrid_train_compact: scratch -> ray-major compact outputs pts [M1,3], density [M1], ray_id / step_id [M1] i64, t [M1].
 * scene_center3 / scene_radius3 are HOST pointers; density_grid is the canonical [P,1,X,Y,Z] parameter. */
int ugrid_train_march(const float *density_grid, int P, int X, int Y, int Z, int freq_num, const float *rays_o,
                      const float *rays_d, int64_t n_rays, const float *t_table, int32_t n_samples,
                      const float *scene_center3, const float *scene_radius3, const float *xyz_min, const float *xyz_max,
                      double bg_len, int norm_l2, float act_shift, float interval, float thres, float *scratch_pts,
                      float *scratch_density, int32_t *scratch_step, int32_t *count, ugrid_stream_t stream);
int ugrid_train_compact(int64_t n_rays, int32_t n_samples, const float *scratch_pts, const float *scratch_density,
                        const int32_t *scratch_step, const int32_t *count, const int64_t *offset_end,
                        const float *t_table, float *pts, float *density, int64_t *ray_id, int64_t *step_id, float *t,
                        ugrid_stream_t stream);

/* Stages 1 AND 2 of the training forward's sampling in one march (FourierGrid_model.py:554-629: ... Raw2Alpha, the alpha mask,
 * Alphas2Weights (render_utils_kernel.cu:598-637), the weight mask and its seven boolean-index gathers).  ugrid_train_sample =
 * ugrid_train_march + the transmittance recurrence over the kept samples, in order, as alpha2weight runs it (same float /
 * double arithmetic, early stop below 1e-3), the weight threshold and alphainv_last [n_rays]; a ray's march ENDS where its
 * transmittance does (the reference evaluates the rest of the ray and discards it: weight 0, gradient 0).  scratch_w /
 * scratch_T: [n_rays * n_samples] like scratch_density; count2 [n_rays] = samples above the weight threshold.  After the caller
 * has prefix-summed count and count2 (int64, inclusive), ugrid_train_sample_compact writes
 *   the M1 = sum(count) stage-1 samples the backward walks: pts1 [M1,3], density1, weights1, T1 [M1], pos2 [M1] (index of the
 *   sample among the stage-2 samples, or -1), and
 *   the M2 = sum(count2) stage-2 samples: pts2 [M2,3], density2, alpha2, weights2, t2 [M2], ray_id2 / step_id2 [M2] int64.
 * ugrid_train_sample_backward: gradients of the stage-2 weights [M2], of alphainv_last [n_rays] and -- added on top -- of the
 * stage-2 raw densities [M2] (any may be NULL) -> g_density1 [M1], the input of ugrid_grid_query_backward on pts1
 * (Alphas2Weights' and Raw2Alpha's backward formulas, render_utils_kernel.cu:639-672, :515-520, evaluated in one pass). */
int ugrid_train_sample(const float *density_grid, int P, int X, int Y, int Z, int freq_num, const float *rays_o,
                       const float *rays_d, int64_t n_rays, const float *t_table, int32_t n_samples,
                       const float *scene_center3, const float *scene_radius3, const float *xyz_min, const float *xyz_max,
                       double bg_len, int norm_l2, float act_shift, float interval, float thres, float *scratch_pts,
                       float *scratch_density, int32_t *scratch_step, float *scratch_w, float *scratch_T, int32_t *count,
                       int32_t *count2, float *alphainv_last, ugrid_stream_t stream);
int ugrid_train_sample_compact(int64_t n_rays, int32_t n_samples, float act_shift, float interval, float thres,
                               const float *scratch_pts, const float *scratch_density, const int32_t *scratch_step,
                               const float *scratch_w, const float *scratch_T, const int32_t *count, const int64_t *offset_end,
                               const int32_t *count2, const int64_t *offset_end2, const float *t_table, float *pts1,
                               float *density1, float *weights1, float *T1, int32_t *pos2, float *pts2, float *density2,
                               float *alpha2, float *weights2, int64_t *ray_id2, int64_t *step_id2, float *t2,
                               ugrid_stream_t stream);
int ugrid_train_sample_backward(int64_t n_rays, float act_shift, float interval, const float *density1, const float *weights1,
                                const float *T1, const int32_t *pos2, const int32_t *count, const int64_t *offset_end,
                                const float *alphainv_last, const float *g_weights2, const float *g_alphainv_last,
                                const float *g_density2, float *g_density1, ugrid_stream_t stream);

/* ugrid_train_sample for the reference's two dense-grid models (NEW, round 4; same scratch / count / compact / backward
 * contract as ugrid_train_sample, density_grid = the canonical [1,1,X,Y,Z] parameter):
 *   ugrid_train_sample_dcvgo  DirectContractedVoxGO.forward's sampling (dcvgo.py:228-330): mid-point table, contraction
 *       ((1 + bg_len) - bg_len / norm, dcvgo.py:258-262), cumdist_thres over the contracted samples (ub360_utils_kernel.cu:13-33;
 *       dist_thres = (2 + 2 bg_len) / world_len * stepsize * 0.95, dcvgo.py:285), maskcache_lookup
 *       (render_utils_kernel.cu:374-392; mask [mi,mj,mk] bytes on the device, mask_dims3 / xyz2ijk_* HOST pointers), dense
 *       trilinear density, Raw2Alpha, alpha mask, Alphas2Weights, weight mask;
 *   ugrid_train_sample_dvgo   DirectVoxGO.forward's (dvgo.py:306-375): sample_pts_on_rays (render_utils_kernel.cu:16-57,
 *       100-260: per-ray t_min / t_max / step count; `far` as the caller passes it, the model passes 1e9), ~mask_outbbox, the mask
 *       cache, then as above.  slots_per_ray = scratch slots per ray, >= the longest possible ray (the box diagonal / stepdist + 2).
 *   ugrid_train_sample_compact_vox = ugrid_train_sample_compact + inner2 [M2] bytes (dcvgo.py:262 inner_mask of the surviving
 *       samples; NULL: not wanted) and t_table == NULL allowed (DirectVoxGO has none: t2 receives float(step_id)). */
int ugrid_train_sample_dcvgo(const float *density_grid, int X, int Y, int Z, const float *rays_o, const float *rays_d,
                             int64_t n_rays, const float *t_table, int32_t n_samples, const float *scene_center3,
                             const float *scene_radius3, const float *xyz_min, const float *xyz_max, double bg_len, int norm_l2,
                             float dist_thres, const uint8_t *mask, const int32_t *mask_dims3, const float *xyz2ijk_scale3,
                             const float *xyz2ijk_shift3, float act_shift, float interval, float thres, float *scratch_pts,
                             float *scratch_density, int32_t *scratch_step, float *scratch_w, float *scratch_T, int32_t *count,
                             int32_t *count2, float *alphainv_last, ugrid_stream_t stream);
int ugrid_train_sample_dvgo(const float *density_grid, int X, int Y, int Z, const float *rays_o, const float *rays_d,
                            int64_t n_rays, int32_t slots_per_ray, const float *xyz_min, const float *xyz_max, float near, float far,
                            float stepdist, const uint8_t *mask, const int32_t *mask_dims3, const float *xyz2ijk_scale3,
                            const float *xyz2ijk_shift3, float act_shift, float interval, float thres, float *scratch_pts,
                            float *scratch_density, int32_t *scratch_step, float *scratch_w, float *scratch_T, int32_t *count,
                            int32_t *count2, float *alphainv_last, ugrid_stream_t stream);
int ugrid_train_sample_compact_vox(int64_t n_rays, int32_t slots_per_ray, float act_shift, float interval, float thres,
                                   const float *scratch_pts, const float *scratch_density, const int32_t *scratch_step,
                                   const float *scratch_w, const float *scratch_T, const int32_t *count, const int64_t *offset_end,
                                   const int32_t *count2, const int64_t *offset_end2, const float *t_table, float *pts1,
                                   float *density1, float *weights1, float *T1, int32_t *pos2, float *pts2, float *density2,
                                   float *alpha2, float *weights2, int64_t *ray_id2, int64_t *step_id2, float *t2,
                                   uint8_t *inner2, ugrid_stream_t stream);

/* NEW (round 5; no reference counterpart): ONE training step's forward and backward of the reference's two dense-grid models
 * issued natively -- DirectVoxGO.forward (dvgo.py:332-425) / DirectContractedVoxGO.forward (dcvgo.py:265-384) /
 * FourierGridModel.forward (FourierGrid_model.py:554-672) with the default
 * 3 x `width` rgbnet (rgbnet_direct), the loss of run_train.py:254-279 and the whole backward, as three calls instead of the
 * ~30 Python-issued launches of the op-by-op step (voxgo_model.py; a 1.2 ms step of which 0.8 ms is GPU time).  The calls run
 * the SAME kernels on the SAME sizes in the same order as that step: results are bit-identical to it.
 *
 *   ugrid_voxgo_step_sample    ugrid_train_sample_dvgo / _dcvgo / ugrid_train_sample, the prefix sums of the two per-ray counts, and the step's ONE
 *                              host read: M1 (stage-1 samples the backward walks) and M2 (samples that reach the rgbnet) are
 *                              written into the struct.  Synchronises `stream` (not with sync_free, below).
 *   (the caller sizes the per-sample buffers: ws = ugrid_voxgo_step_ws_floats floats; the visible outputs below)
 *   ugrid_voxgo_step_forward   ugrid_train_sample_compact(_vox), the k0 lookup, ugrid_rgbnet_features,
 *                              ugrid_rgbnet_train_forward, ugrid_render_loss -> out2 = {loss, mse}, rgb_marched, logits, ...
 *   ugrid_voxgo_step_backward  ugrid_render_loss_backward, ugrid_rgbnet_train_backward (g_w0 .. g_b2 overwritten), the k0
 *                              lookup's scatter into grad_k0_grid (+ touch bitmap when given: channel-last only), and
 *                              ugrid_train_sample_backward + the density scatter into grad_density_grid.  Both grid gradients
 *                              are ADDED to (the caller passes zeros for a fresh gradient).  ws_bwd: ugrid_voxgo_step_bwd_ws_floats.
 * sync_free = 1 (round 6): the step makes NO host read and synchronises nothing -- every call only enqueues work on `stream`, so
 *   a whole step can run ahead of the host or be captured in a hipGraph.  The caller sets M1 / M2 to the CAPACITY of the per-sample
 *   arrays before ugrid_voxgo_step_sample (M1 >= n_rays * slots: the stage-1 arrays cannot overflow; M2 <= M1: stage-2 samples
 *   beyond it are dropped by the compaction -- memory-safe -- and the caller finds totals[1] > M2 when it next looks); the true
 *   counts stay in `totals` on the device, every per-sample kernel of the step processes min(capacity, count) rows in grid-stride
 *   loops whose grids follow hint1 / hint2.  Rows of the per-sample outputs beyond the count are not written.  Forward arrays are
 *   bit-identical to the host-counted step's; the rgbnet's weight gradients are the same sums cut into slabs by the capacity
 *   instead of the count (they differ by rounding when the count is below 32 768).
 * Device pointers unless marked HOST.  mode 0: DirectVoxGO (near / far / stepdist / slots used), 1: DirectContractedVoxGO
 * (t_table[slots] / scene_center / scene_radius / bg_len / norm_l2 / dist_thres used), 2: FourierGridModel
 * (FourierGrid_model.py:509-672: ugrid_train_sample over the Fourier density grid, no mask cache; t_table / scene_center /
 * scene_radius / bg_len / norm_l2 and the two grids' levels used).  bg: [n_rays,3] or NULL.  coef9: see
 * ugrid_render_loss.  inner2 may be NULL.  ugrid_voxgo_step_sizeof = sizeof(ugrid_voxgo_step), for bindings to check their mirror. */
typedef struct ugrid_voxgo_step {
  int32_t mode;
  int32_t k0_channels_last; /* k0_grid / grad_k0_grid stored [P][X][Y][Z][C] */
  int32_t P, freq_num;      /* density grid levels and its fourier_freq_num (mode 2: P = 1 + 2 freq_num; modes 0, 1: 1 and 0) */
  int32_t kP, k0_freq_num;  /* the same for the k0 grid */
  int32_t X, Y, Z;          /* density grid [P,1,X,Y,Z] (canonical) */
  int32_t kX, kY, kZ, C;    /* k0 grid [kP,C,kX,kY,kZ] */
  int32_t pe, width;        /* viewbase_pe; rgbnet width (<= 128; C + 3 + 6 pe <= 128) */
  int32_t slots;            /* scratch slots per ray (mode 0: >= the longest ray's step count; mode 1: the table length) */
  int32_t norm_l2;
  int32_t mask_dims[3];     /* HOST */
  float mask_scale[3], mask_shift[3], scene_center[3], scene_radius[3]; /* HOST */
  float act_shift, interval, thres, near_clip, far_clip, stepdist, dist_thres;
  float coef9[9];           /* HOST */
  double bg_len;
  int64_t n_rays;
  const float *density_grid, *k0_grid, *xyz_min, *xyz_max, *k0_xyz_min, *k0_xyz_max;
  const uint8_t *mask;
  const float *t_table, *viewfreq;
  const float *w0, *b0, *w1, *b1, *w2, *b2;
  const float *rays_o, *rays_d, *viewdirs, *target, *bg;
  /* scratch of the sampling march: [n_rays * slots] (sc_pts: x 3) */
  float *sc_pts, *sc_density;
  int32_t *sc_step;
  float *sc_w, *sc_T;
  int32_t *counts;          /* [2, n_rays] */
  int64_t *offsets;         /* [2, n_rays] inclusive prefix sums of counts */
  int64_t *totals;          /* [2] */
  float *alphainv_last;     /* [n_rays] */
  int64_t *seg;             /* [2 n_rays] */
  float *rgb_marched, *ray_tot, *partial, *out2; /* [n_rays,3], [n_rays,2], [n_rays,5], [2] */
  int64_t M1, M2;           /* written by ugrid_voxgo_step_sample; sync_free: set by the CALLER -- the rows the per-sample arrays hold */
  int64_t hint1, hint2;     /* sync_free: expected counts (e.g. the previous step's), size the launch grids only; 0 = M1 / M2 */
  int32_t sync_free;        /* 1: no host read anywhere in the step (below) */
  int32_t reserved_;
  float *ws;
  float *density2, *alpha2, *weights2, *t2; /* [M2] */
  int64_t *ray_id2, *step_id2;              /* [M2] */
  uint8_t *inner2;                          /* [M2] or NULL */
  float *logits;                            /* [M2,3] */
  /* backward */
  const float *grad_loss;   /* the incoming scalar gradient */
  float *ws_bwd;
  float *g_w0, *g_b0, *g_w1, *g_b1, *g_w2, *g_b2;
  float *grad_density_grid, *grad_k0_grid;
  uint32_t *touch;          /* touched-line bitmap of grad_k0_grid, or NULL */
} ugrid_voxgo_step;
int64_t ugrid_voxgo_step_sizeof(void);
int64_t ugrid_voxgo_step_ws_floats(const ugrid_voxgo_step *s);
int64_t ugrid_voxgo_step_bwd_ws_floats(const ugrid_voxgo_step *s);
int ugrid_voxgo_step_sample(ugrid_voxgo_step *s, ugrid_stream_t stream);
int ugrid_voxgo_step_forward(const ugrid_voxgo_step *s, ugrid_stream_t stream);
int ugrid_voxgo_step_backward(const ugrid_voxgo_step *s, ugrid_stream_t stream);
/* ugrid_voxgo_step_backward in two halves, for a caller that starts the k0 grid's update (the step's largest pass) beside the
 * rest of the backward: _k0 = loss, rgbnet and the k0 scatter (grad_k0_grid complete on the stream; grad_density_grid not needed
 * yet), _density = the sampling's backward + the density scatter.  _backward = the one after the other. */
int ugrid_voxgo_step_backward_k0(const ugrid_voxgo_step *s, ugrid_stream_t stream);
int ugrid_voxgo_step_backward_density(const ugrid_voxgo_step *s, ugrid_stream_t stream);

/* 1 when ugrid_render_shade has an rgbnet instantiation (depth 3, width 128) for this
 * (fourier_freq_num, k0 channels, viewbase_pe) triple, else 0 (they return hipErrorNotSupported for it). */
int ugrid_shade_supported(int32_t freq_num, int32_t k0_channels, int32_t viewbase_pe);

/* Tuning knobs (speed only, never results): "march_waves" 4..6 (waves per SIMD of the march kernel); "tv_xcd" 0|1|2|3
 * (dense TV / TV + Adam kernels: linear workgroup order | XCD-contiguous | + non-temporal streams | + slab order (i-planes of a
 * j-slab stay in L2) for the fused channel-last pass, default 3);
 * "shade_pc" 0|1|2 (shade kernel geometry: classic | 8-wave producer / consumer | 12-wave where it applies, default 2 --
 * bit-identical results).  One knob selects an arithmetic, not only a speed: "train_mlp" 0|1 -- the 33..128-wide products of
 * ugrid_rgbnet_train_forward / _backward on fp32 MFMAs | on bf16x3 (three-way bf16 split, the six part products above 2^-24,
 * fp32 accumulation: fp32-accurate, 2.7 x less matrix time; default 1); results of the two differ in the last bits.
 * Anything else returns hipErrorInvalidValue. */
int ugrid_tune(const char *key, int value);

/* Total survivors of the last march on this ws -> *d_stats (device int64). */
int ugrid_render_stats(void *ws, int64_t n_rays, int32_t n_samples, int64_t *d_stats,
                       ugrid_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UGRID_HIP_H */
