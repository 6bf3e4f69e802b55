// One training step of the two dense-grid models issued natively (include/ugrid_hip.h: ugrid_voxgo_step).
//
// The op-by-op step (voxgo_model.py + train_step.py) is host-bound: ~30 launches through Python + ctypes + four autograd nodes
// take ~0.9 ms to issue for 0.8 ms of GPU time.  Nothing here is a new algorithm: the three entry points call
// the library's own launchers (the same kernels, sizes and order as the Python step -> bit-identical results) and add the two
// things Python did in between -- the prefix sums of the per-ray counts (torch.cumsum) and the read of the two totals.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugrid_common.h"
#include "ugrid_hip.h"

#define ST(s) ((hipStream_t)(s))

thread_local ug_devn ug_tl_devn = {nullptr, 0, 0};      // (ugrid_common.h: the device-resident row count of the sync-free step)

// inclusive int64 prefix sums of the two int32 count rows [2, n] (block b = row b), totals[b] = the row's sum (torch.cumsum of the
// op-by-op step).  Tiles of 1024 threads x 4 consecutive counts: the four loads of a thread are independent (one latency per
// tile, not one per element), thread sums are scanned with wave shuffles + one LDS round over the 16 waves.
#define UG_STEP_SCAN_THREADS 1024
__global__ void __launch_bounds__(UG_STEP_SCAN_THREADS)
k_step_count_scan(const int32_t *__restrict__ counts, int64_t n, int64_t *__restrict__ offsets, int64_t *__restrict__ totals) {
  __shared__ int64_t wave_sum[UG_STEP_SCAN_THREADS / UG_WAVE];
  const int32_t *in = counts + (int64_t)blockIdx.x * n;
  int64_t *out = offsets + (int64_t)blockIdx.x * n;
  const int t = threadIdx.x, lane = t & (UG_WAVE - 1), w = t / UG_WAVE;
  int64_t carry = 0;
  for (int64_t base = 0; base < n; base += 4 * UG_STEP_SCAN_THREADS) {
    const int64_t i0 = base + 4 * (int64_t)t;
    int32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (i0 + j < n) ? in[i0 + j] : 0;
    const int64_t mine = (int64_t)v[0] + v[1] + v[2] + v[3];
    int64_t inc = mine;                                   // inclusive scan of the thread sums within the wave
#pragma unroll
    for (int o = 1; o < UG_WAVE; o <<= 1) {
      const int64_t up = __shfl_up(inc, o, UG_WAVE);
      if (lane >= o) inc += up;
    }
    if (lane == UG_WAVE - 1) wave_sum[w] = inc;
    __syncthreads();
    int64_t before = carry, tile = 0;
#pragma unroll
    for (int k = 0; k < UG_STEP_SCAN_THREADS / UG_WAVE; ++k) {
      const int64_t ws = wave_sum[k];
      if (k < w) before += ws;
      tile += ws;
    }
    int64_t run = before + inc - mine;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      run += v[j];
      if (i0 + j < n) out[i0 + j] = run;
    }
    carry += tile;
    __syncthreads();
  }
  if (t == 0) totals[blockIdx.x] = carry;
}

static inline int64_t ug_al(int64_t floats) { return (floats + 63) & ~(int64_t)63; }      // 256-byte aligned sub-buffers

struct ug_step_ws {      // forward workspace: what the backward needs again, and the rgbnet's input
  float *pts1, *dens1, *w1, *T1;
  int32_t *pos2;
  float *pts2, *k0, *feat, *h1, *h2, *ray_rows;
  int64_t total;
};
struct ug_step_ws_bwd {
  float *g_logits, *g_w, *g_dens, *g_ainv, *g_k0, *g1, *rg;
  int64_t total;
};

static ug_step_ws ug_step_layout(const ugrid_voxgo_step *s) {
  ug_step_ws w;
  const int64_t M1 = s->M1, M2 = s->M2, K = s->C + 3 + 6 * s->pe;
  int64_t o = 0;
  auto take = [&](int64_t n) { float *p = s->ws ? s->ws + o : nullptr; o += ug_al(n); return p; };
  w.pts1 = take(3 * M1); w.dens1 = take(M1); w.w1 = take(M1); w.T1 = take(M1);
  w.pos2 = (int32_t *)take(M1);
  w.pts2 = take(3 * M2); w.k0 = take(M2 * s->C); w.feat = take(M2 * K); w.h1 = take(M2 * s->width); w.h2 = take(M2 * s->width);
  w.ray_rows = take(s->n_rays * (3 + 6 * s->pe));      // the view embedding per ray (ugrid_rgbnet_features' scratch)
  w.total = o;
  return w;
}

static ug_step_ws_bwd ug_step_layout_bwd(const ugrid_voxgo_step *s) {
  ug_step_ws_bwd w;
  const int64_t M1 = s->M1, M2 = s->M2;
  int64_t o = 0;
  auto take = [&](int64_t n) { float *p = s->ws_bwd ? s->ws_bwd + o : nullptr; o += ug_al(n); return p; };
  w.g_logits = take(3 * M2); w.g_w = take(M2); w.g_dens = take(M2); w.g_ainv = take(s->n_rays); w.g_k0 = take(M2 * s->C);
  w.g1 = take(M1); w.rg = take(ugrid_rgbnet_train_scratch_floats(M2));
  w.total = o;
  return w;
}

extern "C" int64_t ugrid_voxgo_step_sizeof(void) { return (int64_t)sizeof(ugrid_voxgo_step); }
extern "C" int64_t ugrid_voxgo_step_ws_floats(const ugrid_voxgo_step *s) { return ug_step_layout(s).total; }
extern "C" int64_t ugrid_voxgo_step_bwd_ws_floats(const ugrid_voxgo_step *s) { return ug_step_layout_bwd(s).total; }

static int ug_step_check(const ugrid_voxgo_step *s) {
  if (!s || s->mode < 0 || s->mode > 2 || s->n_rays <= 0 || s->slots <= 0 || s->C < 1 || s->pe < 0 || s->P < 1 || s->kP < 1 ||
      s->freq_num < 0 || s->k0_freq_num < 0)
    return (int)hipErrorInvalidValue;
  if (s->mode != 2 && (s->P != 1 || s->freq_num != 0)) return (int)hipErrorInvalidValue;       // the dense-grid models' density grid
  if (s->P != 1 + 2 * s->freq_num && !(s->P == 1 && s->freq_num == 0)) return (int)hipErrorInvalidValue;
  if (s->kP != 1 + 2 * s->k0_freq_num && !(s->kP == 1 && s->k0_freq_num == 0)) return (int)hipErrorInvalidValue;
  if (s->width < 1 || s->width > 128 || s->C + 3 + 6 * s->pe > 128) return (int)hipErrorNotSupported;
  if (s->sync_free && (s->M1 < s->n_rays * (int64_t)s->slots || s->M2 < 1 || s->M2 > s->M1 || s->hint1 < 0 || s->hint2 < 0))
    return (int)hipErrorInvalidValue;                    // capacities: stage 1 cannot overflow, stage 2 is clamped by the compaction
  return 0;
}
// the count of the step's stage-1 / stage-2 arrays as the kernels see it: on the device (sync_free) or the host's number
#define UG_STEP_ROWS1(s) ug_devn_scope rows_scope_((s)->sync_free ? (s)->totals : nullptr, (s)->hint1, 0)
#define UG_STEP_ROWS2(s) ug_devn_scope rows_scope_((s)->sync_free ? (s)->totals + 1 : nullptr, (s)->hint2, (s)->sync_free ? (s)->M2 : 0)

extern "C" int ugrid_voxgo_step_sample(ugrid_voxgo_step *s, ugrid_stream_t st) {
  int rc = ug_step_check(s);
  if (rc) return rc;
  const int64_t R = s->n_rays;
  int32_t *c1 = s->counts, *c2 = s->counts + R;
  if (s->mode == 2)
    rc = ugrid_train_sample(s->density_grid, s->P, s->X, s->Y, s->Z, s->freq_num, s->rays_o, s->rays_d, R, s->t_table, s->slots,
                            s->scene_center, s->scene_radius, s->xyz_min, s->xyz_max, s->bg_len, s->norm_l2, s->act_shift, s->interval,
                            s->thres, s->sc_pts, s->sc_density, s->sc_step, s->sc_w, s->sc_T, c1, c2, s->alphainv_last, st);
  else if (s->mode == 1)
    rc = ugrid_train_sample_dcvgo(s->density_grid, s->X, s->Y, s->Z, s->rays_o, s->rays_d, R, s->t_table, s->slots, s->scene_center,
                                  s->scene_radius, s->xyz_min, s->xyz_max, s->bg_len, s->norm_l2, s->dist_thres, s->mask, s->mask_dims,
                                  s->mask_scale, s->mask_shift, s->act_shift, s->interval, s->thres, s->sc_pts, s->sc_density, s->sc_step,
                                  s->sc_w, s->sc_T, c1, c2, s->alphainv_last, st);
  else
    rc = ugrid_train_sample_dvgo(s->density_grid, s->X, s->Y, s->Z, s->rays_o, s->rays_d, R, s->slots, s->xyz_min, s->xyz_max, s->near_clip,
                                 s->far_clip, s->stepdist, s->mask, s->mask_dims, s->mask_scale, s->mask_shift, s->act_shift, s->interval,
                                 s->thres, s->sc_pts, s->sc_density, s->sc_step, s->sc_w, s->sc_T, c1, c2, s->alphainv_last, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_step_count_scan, dim3(2), dim3(UG_STEP_SCAN_THREADS), 0, ST(st), s->counts, R, s->offsets, s->totals);
  UG_LAUNCH_CHECK();
  if (s->sync_free) return 0;                          // M1 / M2 stay the caller's capacities; the counts stay in s->totals
  static thread_local int64_t *pinned = nullptr;      // the step's one host read lands in page-locked memory
  if (!pinned) UG_HIP(hipHostMalloc((void **)&pinned, 2 * sizeof(int64_t), hipHostMallocDefault));
  UG_HIP(hipMemcpyAsync(pinned, s->totals, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, ST(st)));
  UG_HIP(hipStreamSynchronize(ST(st)));
  s->M1 = pinned[0];
  s->M2 = pinned[1];
  return 0;
}

extern "C" int ugrid_voxgo_step_forward(const ugrid_voxgo_step *s, ugrid_stream_t st) {
  int rc = ug_step_check(s);
  if (rc) return rc;
  if (s->M1 < 0 || s->M2 < 0 || s->M2 > s->M1 || ((s->M1 | s->M2) && !s->ws)) return (int)hipErrorInvalidValue;
  const ug_step_ws w = ug_step_layout(s);
  const int64_t R = s->n_rays, M2 = s->M2;
  const int K = s->C + 3 + 6 * s->pe;
  UG_STEP_ROWS2(s);
  if (s->M1 > 0 && s->mode == 2) {
    rc = ugrid_train_sample_compact(R, s->slots, s->act_shift, s->interval, s->thres, s->sc_pts, s->sc_density, s->sc_step, s->sc_w, s->sc_T,
                                    s->counts, s->offsets, s->counts + R, s->offsets + R, s->t_table, w.pts1, w.dens1, w.w1, w.T1, w.pos2,
                                    w.pts2, s->density2, s->alpha2, s->weights2, s->ray_id2, s->step_id2, s->t2, st);
    if (rc) return rc;
  } else if (s->M1 > 0) {
    rc = ugrid_train_sample_compact_vox(R, s->slots, s->act_shift, s->interval, s->thres, s->sc_pts, s->sc_density, s->sc_step, s->sc_w,
                                        s->sc_T, s->counts, s->offsets, s->counts + R, s->offsets + R, s->mode == 1 ? s->t_table : nullptr,
                                        w.pts1, w.dens1, w.w1, w.T1, w.pos2, w.pts2, s->density2, s->alpha2, s->weights2, s->ray_id2,
                                        s->step_id2, s->t2, s->mode == 1 ? s->inner2 : nullptr, st);
    if (rc) return rc;
  }
  rc = (s->k0_channels_last ? ugrid_grid_query_cl : ugrid_grid_query)(s->k0_grid, s->kP, s->C, s->kX, s->kY, s->kZ, w.pts2, s->k0_xyz_min,
                                                                     s->k0_xyz_max, s->k0_freq_num, M2, w.k0, st);
  if (rc) return rc;
  rc = ugrid_rgbnet_features(w.k0, s->C, s->viewdirs, R, s->viewfreq, s->pe, s->ray_id2, M2, w.ray_rows, w.feat, st);
  if (rc) return rc;
  rc = ugrid_rgbnet_train_forward(w.feat, M2, K, s->w0, s->b0, s->w1, s->b1, s->w2, s->b2, s->width, w.h1, w.h2, s->logits, st);
  if (rc) return rc;
  return ugrid_render_loss(s->logits, s->weights2, nullptr, s->t2, s->alphainv_last, s->bg, s->target, s->ray_id2, M2, R, s->coef9, s->seg,
                           s->rgb_marched, s->ray_tot, s->partial, s->out2, st);
}

// first half of the backward: everything up to the k0 grid's gradient (complete when this returns to the stream)
extern "C" int ugrid_voxgo_step_backward_k0(const ugrid_voxgo_step *s, ugrid_stream_t st) {
  int rc = ug_step_check(s);
  if (rc) return rc;
  if (!s->grad_loss || !s->ws_bwd || !s->grad_k0_grid) return (int)hipErrorInvalidValue;
  const ug_step_ws w = ug_step_layout(s);
  const ug_step_ws_bwd b = ug_step_layout_bwd(s);
  const int64_t R = s->n_rays, M2 = s->M2;
  const int K = s->C + 3 + 6 * s->pe;
  UG_STEP_ROWS2(s);
  rc = ugrid_render_loss_backward(s->logits, s->weights2, nullptr, s->t2, s->alphainv_last, s->bg, s->target, s->ray_id2, M2, R, s->coef9,
                                  s->seg, s->rgb_marched, s->ray_tot, s->grad_loss, b.g_logits, b.g_w, b.g_ainv, b.g_dens, st);
  if (rc) return rc;
  rc = ugrid_rgbnet_train_backward(b.g_logits, w.feat, w.h1, w.h2, M2, K, s->C, s->w0, s->w1, s->w2, s->width, b.g_k0, s->g_w0, s->g_b0,
                                   s->g_w1, s->g_b1, s->g_w2, s->g_b2, b.rg, st);
  if (rc) return rc;
  if (s->k0_channels_last && s->touch)
    return ugrid_grid_query_backward_cl_touch(b.g_k0, s->kP, s->C, s->kX, s->kY, s->kZ, w.pts2, s->k0_xyz_min, s->k0_xyz_max, s->k0_freq_num,
                                              M2, s->grad_k0_grid, s->touch, st);
  return (s->k0_channels_last ? ugrid_grid_query_backward_cl : ugrid_grid_query_backward)(
      b.g_k0, s->kP, s->C, s->kX, s->kY, s->kZ, w.pts2, s->k0_xyz_min, s->k0_xyz_max, s->k0_freq_num, M2, s->grad_k0_grid, st);
}

// second half: the sampling's backward and the density grid's gradient (reads what the first half left in ws_bwd)
extern "C" int ugrid_voxgo_step_backward_density(const ugrid_voxgo_step *s, ugrid_stream_t st) {
  int rc = ug_step_check(s);
  if (rc) return rc;
  if (!s->ws_bwd || !s->grad_density_grid) return (int)hipErrorInvalidValue;
  if (s->M1 <= 0) return 0;
  const ug_step_ws w = ug_step_layout(s);
  const ug_step_ws_bwd b = ug_step_layout_bwd(s);
  UG_STEP_ROWS1(s);
  rc = ugrid_train_sample_backward(s->n_rays, s->act_shift, s->interval, w.dens1, w.w1, w.T1, w.pos2, s->counts, s->offsets, s->alphainv_last,
                                   b.g_w, b.g_ainv, b.g_dens, b.g1, st);
  if (rc) return rc;
  return ugrid_grid_query_backward(b.g1, s->P, 1, s->X, s->Y, s->Z, w.pts1, s->xyz_min, s->xyz_max, s->freq_num, s->M1, s->grad_density_grid,
                                   st);
}

extern "C" int ugrid_voxgo_step_backward(const ugrid_voxgo_step *s, ugrid_stream_t st) {
  const int rc = ugrid_voxgo_step_backward_k0(s, st);
  return rc ? rc : ugrid_voxgo_step_backward_density(s, st);
}
