// libugrid_hip.so -- march half of the fused render path + grid packing/query kernels.
// Built with -fno-slp-vectorize (see ugrid_render.h header note and csrc/build.sh).
#include "ugrid_render.h"

// ----------------------------------------------------------------------------------------------
// brick packing: canonical [P,C,X,Y,Z] -> [P*(X-1)(Y-1)(Z-1)] cell records of [H halves][8][CH]
// coef=1 (density and feature bricks): the 8 entries of a cell are the coefficients of its trilinear
// polynomial  f(tx,ty,tz) = sum_{dx,dy,dz} c[4dx+2dy+dz] tx^dx ty^dy tz^dz  (t = fractional cell coordinates),
// computed in fp64 from the 8 corner values and rounded once: the kernels evaluate a cell with 7 FMAs per
// channel and need no corner weights.  coef=0 (rgbnet-less colour bricks): the 8 corner values themselves.
// ----------------------------------------------------------------------------------------------
__global__ void k_pack_bricks(const float *__restrict__ grid, int P, int C, int X, int Y, int Z, int H,
                              int CH, int coef, float *__restrict__ out, int64_t total) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    int64_t q = o;
    int ch, c;
    if (H == 2) {   // feature half-bricks: [pair CH/2][entry 8][2 channels]
      const int qi = (int)(q % (8 * CH)); q /= (8 * CH);
      c = (qi >> 1) & 7;
      ch = (qi >> 4) * 2 + (qi & 1);
    } else {        // density / rgbnet-less colour bricks: [entry 8][CH]
      ch = (int)(q % CH); q /= CH;
      c = (int)(q % 8); q /= 8;
    }
    const int h = (int)(q % H); q /= H;
    const int k = (int)(q % (Z - 1)); q /= (Z - 1);
    const int j = (int)(q % (Y - 1)); q /= (Y - 1);
    const int i = (int)(q % (X - 1)); q /= (X - 1);
    const int l = (int)q;
    const int chan = h * CH + ch;
    float v = 0.f;
    if (chan < C) {
      const float *g = grid + ((int64_t)l * C + chan) * X * Y * Z;
      if (!coef) {
        const int ii = i + (c >> 2), jj = j + ((c >> 1) & 1), kk = k + (c & 1);
        v = g[((int64_t)ii * Y + jj) * Z + kk];
      } else {
        // inclusion-exclusion over the corners s that are sub-masks of c
        double acc = 0.0;
        for (int s = 0; s < 8; ++s) {
          if (s & ~c) continue;
          const int ii = i + (s >> 2), jj = j + ((s >> 1) & 1), kk = k + (s & 1);
          const double t = (double)g[((int64_t)ii * Y + jj) * Z + kk];
          acc += (__popc(c ^ s) & 1) ? -t : t;
        }
        v = (float)acc;
      }
    }
    out[o] = v;
  }
}

// quad layout of the C == 12 feature bricks (ugrid_render.h "quad k0 gather"): [cell][q 0..5][g 0..3][4 floats]; lane g
// of a quad owns channels 3g..3g+2, float4 q holds coefficients 4(q&1)..4(q&1)+3 of channel 3g + (q>>1)
__global__ void k_pack_quad(const float *__restrict__ grid, int P, int C, int X, int Y, int Z, float *__restrict__ out,
                            int64_t total) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(o & 3), g = (int)((o >> 2) & 3);
    int64_t r = o >> 4;
    const int q = (int)(r % 6); r /= 6;
    const int k = (int)(r % (Z - 1)); r /= (Z - 1);
    const int j = (int)(r % (Y - 1)); r /= (Y - 1);
    const int i = (int)(r % (X - 1)); r /= (X - 1);
    const int l = (int)r;
    const int chan = 3 * g + (q >> 1), c = 4 * (q & 1) + e;
    const float *gp = grid + ((int64_t)l * C + chan) * X * Y * Z;
    double acc = 0.0;   // inclusion-exclusion over the corners s that are sub-masks of c (as k_pack_bricks)
    for (int s = 0; s < 8; ++s) {
      if (s & ~c) continue;
      const int ii = i + (s >> 2), jj = j + ((s >> 1) & 1), kk = k + (s & 1);
      const double t = (double)gp[((int64_t)ii * Y + jj) * Z + kk];
      acc += (__popc(c ^ s) & 1) ? -t : t;
    }
    out[o] = (float)acc;
  }
}

// ----------------------------------------------------------------------------------------------
// stand-alone grid query on the canonical layout (FourierGrid.forward / DenseGrid.forward).
// 1 lane per point; corner taps are z-pairs in the [.., Z] fastest dimension.
// ----------------------------------------------------------------------------------------------
// generic zero-padded trilinear tap at normalised (cx->X axis, cy->Y, cz->Z): the 8 corner offsets inside one
// [X,Y,Z] plane (-1 = outside the grid, zero padding) and their weights, in grid_sample's corner order
struct ug_taps { int64_t off[8]; float w[8]; };
__device__ __forceinline__ ug_taps ug_tap_setup(int X, int Y, int Z, float cx, float cy, float cz) {
  const float ix = ((cx + 1.f) / 2.f) * (float)(X - 1);
  const float iy = ((cy + 1.f) / 2.f) * (float)(Y - 1);
  const float iz = ((cz + 1.f) / 2.f) * (float)(Z - 1);
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const float wx0 = (fx + 1.f) - ix, wx1 = ix - fx;
  const float wy0 = (fy + 1.f) - iy, wy1 = iy - fy;
  const float wz0 = (fz + 1.f) - iz, wz1 = iz - fz;
  const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)X), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Y),
            z0 = (int)fminf(fmaxf(fz, -2.f), (float)Z);
  ug_taps t;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int xi = x0 + (c >> 2), yi = y0 + ((c >> 1) & 1), zi = z0 + (c & 1);
    const bool in = xi >= 0 && xi < X && yi >= 0 && yi < Y && zi >= 0 && zi < Z;
    t.off[c] = in ? ((int64_t)xi * Y + yi) * Z + zi : -1;
    t.w[c] = ((c & 1) ? wz1 : wz0) * (((c >> 1) & 1) ? wy1 : wy0) * ((c >> 2) ? wx1 : wx0);
  }
  return t;
}

// The same taps for kernels that LOAD the corners: every offset is valid (a corner outside the grid is clamped, per axis, onto the cell's
// in-range corner) and its weight is 0 -- `acc += g[off] * w` then runs without a branch and adds an exact zero where grid_sample pads.
// With `if (off >= 0) acc += g[off] * w` every load sat under its own exec branch and hipcc put s_waitcnt vmcnt(0) behind each: the eight
// corners of a level fetched ONE AFTER THE OTHER (round 6, visit P; S3's sampling march 0.64 ms).  Bit-identical: x + 0 = x, and the
// clamped corner is one the sample reads anyway (a NaN there reaches the result in both forms).
__device__ __forceinline__ ug_taps ug_tap_setup_ld(int X, int Y, int Z, float cx, float cy, float cz) {
  const float ix = ((cx + 1.f) / 2.f) * (float)(X - 1);
  const float iy = ((cy + 1.f) / 2.f) * (float)(Y - 1);
  const float iz = ((cz + 1.f) / 2.f) * (float)(Z - 1);
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const float wx0 = (fx + 1.f) - ix, wx1 = ix - fx;
  const float wy0 = (fy + 1.f) - iy, wy1 = iy - fy;
  const float wz0 = (fz + 1.f) - iz, wz1 = iz - fz;
  const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)X), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Y),
            z0 = (int)fminf(fmaxf(fz, -2.f), (float)Z);
  ug_taps t;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int xi = x0 + (c >> 2), yi = y0 + ((c >> 1) & 1), zi = z0 + (c & 1);
    const bool in = xi >= 0 && xi < X && yi >= 0 && yi < Y && zi >= 0 && zi < Z;
    const int xc = min(max(xi, 0), X - 1), yc = min(max(yi, 0), Y - 1), zc = min(max(zi, 0), Z - 1);
    t.off[c] = ((int64_t)xc * Y + yc) * Z + zc;
    const float w = ((c & 1) ? wz1 : wz0) * (((c >> 1) & 1) ? wy1 : wy0) * ((c >> 2) ? wx1 : wx0);
    t.w[c] = in ? w : 0.f;
  }
  return t;
}

// level coordinates of the Fourier grid: l = 0 plain, odd l = sin(2^k u), even l = cos(2^k u), k = (l-1)/2
__device__ __forceinline__ void ug_level_coords(int l, float ux, float uy, float uz, float &cx, float &cy, float &cz) {
  cx = ux; cy = uy; cz = uz;
  if (l > 0) {
    // ug_sincos (ugrid_math.h): 1.2e-7 max abs error for |x| <= 4, i.e. within an ulp of libm's results at a
    // seventh of the instructions of ocml sinf / cosf
    const float f = (float)(1 << ((l - 1) >> 1));
    float sx, kx, sy, ky, sz, kz;
    ug_sincos(f * ux, &sx, &kx);
    ug_sincos(f * uy, &sy, &ky);
    ug_sincos(f * uz, &sz, &kz);
    const bool is_cos = (l - 1) & 1;
    cx = is_cos ? kx : sx; cy = is_cos ? ky : sy; cz = is_cos ? kz : sz;
  }
}

// forward: 1 lane per point; per level the taps are set up once and reused by every channel; the per-channel sums
// over levels build up in the output row itself (level 0 first, like the reference's mean over the level axis).
// CL = channel-last storage [P][X][Y][Z][C] (torch channels_last_3d of the same logical [P,C,X,Y,Z] tensor: the C
// channels of a voxel are one contiguous run -- the training layout of multi-channel grids, section 4.4 of DESIGN.md);
// the arithmetic and its order are the same in both layouts.
template <bool CL>
__device__ __forceinline__ void ug_grid_query_one(int64_t tid, const float *__restrict__ grid, int P, int C, int X, int Y, int Z,
                                                  const float *__restrict__ xyz, const float *__restrict__ xyz_min,
                                                  const float *__restrict__ xyz_max, int F, int64_t n, float *__restrict__ out) {
  // canonical layout: one lane per point, levels and channels in loops.  channel-last: one lane per (point, channel),
  // channel fastest -- the C lanes of a point read one 4C-byte voxel record per corner; same per-channel operation order
  const int64_t p = CL ? tid / C : tid;
  if (p >= n) return;
  const float ux = ug_unorm(xyz[3 * p], xyz_min[0], xyz_max[0]);
  const float uy = ug_unorm(xyz[3 * p + 1], xyz_min[1], xyz_max[1]);
  const float uz = ug_unorm(xyz[3 * p + 2], xyz_min[2], xyz_max[2]);
  const int64_t vol = (int64_t)X * Y * Z;
  if (CL) {
    const int ch = (int)(tid - p * C);
    auto level = [&](int l, float cx, float cy, float cz) -> float {
      const ug_taps t = ug_tap_setup_ld(X, Y, Z, cx, cy, cz);
      const float *__restrict__ g = grid + (int64_t)l * vol * C + ch;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc += g[t.off[c] * C] * t.w[c];
      return acc;
    };
    // level 0, then the sin and the cos level of every frequency TOGETHER: one sin / cos evaluation serves both (ug_level_coords formed it
    // once per level) and their 16 corner loads are in flight at once; the level sums are added in level order as before: bit-identical
    float sum = level(0, ux, uy, uz);
    for (int k = 0; 2 * k + 2 < P; ++k) {
      const float f = (float)(1 << k);
      float sx, kx, sy, ky, sz, kz;
      ug_sincos(f * ux, &sx, &kx);
      ug_sincos(f * uy, &sy, &ky);
      ug_sincos(f * uz, &sz, &kz);
      const ug_taps ts = ug_tap_setup_ld(X, Y, Z, sx, sy, sz), tc = ug_tap_setup_ld(X, Y, Z, kx, ky, kz);
      const float *__restrict__ gs = grid + (int64_t)(2 * k + 1) * vol * C + ch, *__restrict__ gc = grid + (int64_t)(2 * k + 2) * vol * C + ch;
      float vs[8], vc[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) vs[c] = gs[ts.off[c] * C];
#pragma unroll
      for (int c = 0; c < 8; ++c) vc[c] = gc[tc.off[c] * C];
      float as = 0.f, ac = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) as += vs[c] * ts.w[c];
#pragma unroll
      for (int c = 0; c < 8; ++c) ac += vc[c] * tc.w[c];
      sum = sum + as;
      sum = sum + ac;
    }
    out[tid] = (F > 0) ? sum / (float)P : sum;
    return;
  }
  float *__restrict__ row = out + p * C;
  {
    const ug_taps t = ug_tap_setup_ld(X, Y, Z, ux, uy, uz);
    for (int ch = 0; ch < C; ++ch) {
      const float *__restrict__ g = grid + (int64_t)ch * vol;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc += g[t.off[c]] * t.w[c];
      row[ch] = acc;
    }
  }
  for (int k = 0; 2 * k + 2 < P; ++k) {          // the sin and the cos level of a frequency together (one sin / cos evaluation, 16 loads in flight)
    const float f = (float)(1 << k);
    float sx, kx, sy, ky, sz, kz;
    ug_sincos(f * ux, &sx, &kx);
    ug_sincos(f * uy, &sy, &ky);
    ug_sincos(f * uz, &sz, &kz);
#pragma unroll
    for (int half = 0; half < 2; ++half) {          // the sin level, then the cos level (one sin / cos evaluation serves both)
      const ug_taps t = half == 0 ? ug_tap_setup_ld(X, Y, Z, sx, sy, sz) : ug_tap_setup_ld(X, Y, Z, kx, ky, kz);
      for (int ch = 0; ch < C; ++ch) {
        const float *__restrict__ g = grid + ((int64_t)(2 * k + 1 + half) * C + ch) * vol;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += g[t.off[c]] * t.w[c];
        row[ch] = row[ch] + acc;
      }
    }
  }
  if (F > 0)
    for (int ch = 0; ch < C; ++ch) row[ch] = row[ch] / (float)P;
}
// n_dev (may be null): the row count on the device (ug_devn, ugrid_common.h); grid-stride, so that a grid sized by a hint covers any count
template <bool CL>
__global__ void k_grid_query(const float *__restrict__ grid, int P, int C, int X, int Y, int Z,
                             const float *__restrict__ xyz, const float *__restrict__ xyz_min,
                             const float *__restrict__ xyz_max, int F, int64_t n, float *__restrict__ out,
                             const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(n, n_dev);
  const int64_t total = CL ? n * C : n;
  for (int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += (int64_t)gridDim.x * blockDim.x)
    ug_grid_query_one<CL>(tid, grid, P, C, X, Y, Z, xyz, xyz_min, xyz_max, F, n, out);
}

// backward w.r.t. the grid: 1 lane per (point, level); grad_grid[l, ch, corner] += w * grad_out[p, ch] / P with
// hardware fp32 atomics (global_atomic_add_f32).  Like torch's grid_sample backward the accumulation order is
// not fixed, so sums agree to rounding, not bit for bit.  Canonical layout: the C atomics of a corner land in C
// different planes (32 MB apart at G = 200) -- every atomic its own 128-byte line; channel-last: one 4C-byte run.
// `touch` (channel-last only, may be null): one bit per 256-byte line (64 floats) of grad_grid, set for every line this
// launch adds to -- the masked TV / Adam passes then visit only those lines instead of scanning the whole gradient for
// non-zeros (ugrid_touch_words; the marking happens before the lane's own zero test, so that a record any channel of which
// receives a gradient is always marked).
template <bool CL>
__device__ __forceinline__ void ug_grid_query_backward_one(int64_t tid, const float *__restrict__ grad_out, int P, int C, int X, int Y, int Z,
                                                           const float *__restrict__ xyz, const float *__restrict__ xyz_min,
                                                           const float *__restrict__ xyz_max, int F, int64_t n,
                                                           float *__restrict__ grad_grid, uint32_t *__restrict__ touch) {
  // canonical layout: one lane per (point, level), channels in a loop (each channel is its own volume).
  // channel-last layout: one lane per (point, level, channel), channel fastest -- the C lanes of one (point, level) hit
  // C consecutive floats of one voxel record, so each atomic instruction touches ~64/C records instead of 64 lines.
  const int64_t q = CL ? tid / C : tid;
  const int chl = CL ? (int)(tid - q * C) : 0;
  if (q >= n * P) return;
  const int64_t p = q / P;
  const int l = (int)(q - p * P);
  const float ux = ug_unorm(xyz[3 * p], xyz_min[0], xyz_max[0]);
  const float uy = ug_unorm(xyz[3 * p + 1], xyz_min[1], xyz_max[1]);
  const float uz = ug_unorm(xyz[3 * p + 2], xyz_min[2], xyz_max[2]);
  float cx, cy, cz;
  ug_level_coords(l, ux, uy, uz, cx, cy, cz);
  const ug_taps t = ug_tap_setup(X, Y, Z, cx, cy, cz);
  const int64_t vol = (int64_t)X * Y * Z;
  if (CL) {
    if (touch) {
      // the two z-neighbours of a corner pair are adjacent records: one run of <= 2C floats, marked by one lane
      const int64_t e0 = (int64_t)l * vol * C;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if ((C >= 4 ? k : 0) != chl) continue;          // pair k is lane k's (a record with < 4 channels: all lane 0's)
        const int64_t lo = t.off[2 * k] >= 0 ? t.off[2 * k] : t.off[2 * k + 1];
        const int64_t hi = t.off[2 * k + 1] >= 0 ? t.off[2 * k + 1] : t.off[2 * k];
        if (lo < 0) continue;
        for (int64_t line = (e0 + lo * C) >> 6; line <= (e0 + hi * C + C - 1) >> 6; ++line) {
          const uint32_t bit = 1u << (line & 31);
          if (!(touch[line >> 5] & bit)) atomicOr(touch + (line >> 5), bit);
        }
      }
    }
    float g = grad_out[p * C + chl];
    if (F > 0) g = g / (float)P;
    if (g == 0.f) return;     // exact zeros stay exact zeros in the grid gradient (MaskedAdam keys on them)
    float *__restrict__ gl = grad_grid + (int64_t)l * vol * C + chl;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (t.off[c] >= 0) unsafeAtomicAdd(gl + t.off[c] * C, g * t.w[c]);
    return;
  }
  for (int ch = 0; ch < C; ++ch) {
    float g = grad_out[p * C + ch];
    if (F > 0) g = g / (float)P;
    if (g == 0.f) continue;   // exact zeros stay exact zeros in the grid gradient (MaskedAdam keys on them)
    float *__restrict__ gg = grad_grid + ((int64_t)l * C + ch) * vol;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (t.off[c] >= 0) unsafeAtomicAdd(gg + t.off[c], g * t.w[c]);
  }
}
template <bool CL>
__global__ void k_grid_query_backward(const float *__restrict__ grad_out, int P, int C, int X, int Y, int Z,
                                      const float *__restrict__ xyz, const float *__restrict__ xyz_min,
                                      const float *__restrict__ xyz_max, int F, int64_t n,
                                      float *__restrict__ grad_grid, uint32_t *__restrict__ touch, const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(n, n_dev);
  const int64_t total = (CL ? n * C : n) * P;
  for (int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += (int64_t)gridDim.x * blockDim.x)
    ug_grid_query_backward_one<CL>(tid, grad_out, P, C, X, Y, Z, xyz, xyz_min, xyz_max, F, n, grad_grid, touch);
}

// ----------------------------------------------------------------------------------------------
// Training forward, stage 1 (new entry points; replace the head of FourierGridModel.forward in training mode,
// FourierGrid_model.py:554-598: sample_ray -> [R,S,3] points, density lookup on all R*S points, Raw2Alpha, the
// `alpha > fast_color_thres` mask and seven boolean-index gathers with their host syncs).
//   k_train_march   one wave per ray, lane = sample (S/64 rounds): point, contraction, P-level density on the CANONICAL
//                   layout (the parameter being trained; same device functions and operation order as k_grid_query,
//                   so the densities are bit-identical to the composed path), alpha exactly as k_raw2alpha forms it,
//                   threshold, ballot + mbcnt compaction into the ray's private slot [r*S, r*S + count[r]) of a scratch
//   k_train_compact after a cumsum of the counts: one wave per ray copies its slot to the ray-major compact outputs
// The point arithmetic follows the torch elementwise chain of sample_ray (separate multiply / add, IEEE divisions).
// ----------------------------------------------------------------------------------------------
struct ug_train_args {
  int64_t n_rays;
  int32_t S, P, F, X, Y, Z, norm_l2;
  float cx, cy, cz, rx, ry, rz;
  float B, A;
  float shift, interval, thres;
};

// MODE 1 / 2 of k_train_march: the sampling rules of the reference's two non-Fourier models (single-level dense grids)
//   1 = DirectContractedVoxGO (dcvgo.py:228-310): the FourierGrid point rule + of the contracted samples only those whose running
//       inter-sample distance has just exceeded dist_thres (cumdist_thres, ub360_utils_kernel.cu:13-33: a serial recurrence
//       along the ray -- here a readlane loop over the wave's 64 samples with a wave-uniform carry) + the mask cache
//       (maskcache_lookup, render_utils_kernel.cu:374-392); the step id of a contracted sample is stored complemented
//       (inner_mask of dcvgo.py:262 travels in the sign)
//   2 = DirectVoxGO (dvgo.py:306-400): per-ray box clipping and step count (infer_t_minmax / infer_n_samples /
//       sample_pts_on_rays, render_utils_kernel.cu:16-57,100-260), mask_outbbox, the mask cache; a.S = slots per ray (>= the
//       longest possible ray: the host sizes it for the box diagonal), t_table unused
struct ug_train_vox {
  const uint8_t *mask;
  int32_t mi, mj, mk;
  float sx, sy, sz, hx, hy, hz;     // xyz2ijk_scale / xyz2ijk_shift
  float dist_thres;                 // mode 1
  float near, far, stepdist;        // mode 2
};

// mask cache: nearest voxel, C round(), NaN -> 0 like the device conversion (k_maskcache)
__device__ __forceinline__ bool ug_train_maskcache(const ug_train_vox &v, float px, float py, float pz) {
  float fi = roundf(px * v.sx + v.hx), fj = roundf(py * v.sy + v.hy), fk = roundf(pz * v.sz + v.hz);
  fi = (fi != fi) ? 0.f : fi; fj = (fj != fj) ? 0.f : fj; fk = (fk != fk) ? 0.f : fk;
  if (fi >= 0.f && fi < (float)v.mi && fj >= 0.f && fj < (float)v.mj && fk >= 0.f && fk < (float)v.mk)
    return v.mask[((int64_t)fi * v.mj + (int64_t)fj) * v.mk + (int64_t)fk] != 0;
  return false;
}

// sample t of a normalised ray, contracted outside the unit cube / ball (dcvgo.py:251-262, the arithmetic of k_train_march);
// returns the norm before the contraction
__device__ __forceinline__ float ug_train_point(const ug_train_args &a, float ox, float oy, float oz, float dx, float dy, float dz,
                                                float t, float &px, float &py, float &pz) {
  px = ox + dx * t; py = oy + dy * t; pz = oz + dz * t;
  const float nrm = a.norm_l2 ? ug_norm3_torch(px, py, pz) : fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz)));
  if (!(nrm <= 1.0f)) {
    // `bg_len / norm` with a Python number on the left is torch's Tensor.__rtruediv__ = reciprocal(norm) * bg_len: two roundings
    const float sc = a.B - (1.0f / nrm) * a.A;
    px = px / nrm * sc; py = py / nrm * sc; pz = pz / nrm * sc;
  }
  return nrm;
}

// alpha of a raw density exactly as k_raw2alpha forms it (e = expf(density + shift) is what its backward keeps)
__device__ __forceinline__ float ug_train_alpha(float dens, float shift, float interval, float *e_out) {
  const float e = expf(dens + shift);
  *e_out = e;
  return 1 - powf(1 + e, -interval);
}

// W2 = true (ugrid_train_sample): ALSO stage 2 of the sampling -- the transmittance recurrence of Alphas2Weights over the
// kept samples, in order, exactly as k_alpha2weight runs it (T in float, the product in double, early stop below 1e-3),
// the weight threshold, alphainv_last -- and the march of a ray ENDS where its transmittance does (the reference looks up
// all S samples and throws the tail away: weight 0, gradient 0).  s_w / s_T: weight and transmittance per kept sample,
// count2: kept samples whose weight exceeds the threshold.
template <bool W2>
__global__ void __launch_bounds__(256)
k_train_march(ug_train_args a, const float *__restrict__ grid, const float *__restrict__ rays_o,
              const float *__restrict__ rays_d, const float *__restrict__ t_table, const float *__restrict__ xyz_min,
              const float *__restrict__ xyz_max, float *__restrict__ s_pts, float *__restrict__ s_dens,
              int32_t *__restrict__ s_step, int32_t *__restrict__ count, float *__restrict__ s_w, float *__restrict__ s_T,
              int32_t *__restrict__ count2, float *__restrict__ alphainv_last) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ray >= a.n_rays) return;
  const int lane = ug_lane();
  const float ox = (rays_o[3 * ray] - a.cx) / a.rx, oy = (rays_o[3 * ray + 1] - a.cy) / a.ry, oz = (rays_o[3 * ray + 2] - a.cz) / a.rz;
  const float rdx = rays_d[3 * ray], rdy = rays_d[3 * ray + 1], rdz = rays_d[3 * ray + 2];
  const float dn = ug_norm3_torch(rdx, rdy, rdz);
  const float dx = rdx / dn, dy = rdy / dn, dz = rdz / dn;
  const float lox = xyz_min[0], loy = xyz_min[1], loz = xyz_min[2], hix = xyz_max[0], hiy = xyz_max[1], hiz = xyz_max[2];
  const int64_t vol = (int64_t)a.X * a.Y * a.Z;
  const int64_t slot = ray * a.S;
  int kept = 0, kept2 = 0;   // wave-uniform
  float T_cum = 1.f;         // wave-uniform
  bool stopped = false;
  for (int j0 = 0; j0 < a.S && !stopped; j0 += UG_WAVE) {
    const int j = j0 + lane;
    bool keep = false;
    float px = 0.f, py = 0.f, pz = 0.f, dens = 0.f, alpha = 0.f;
    if (j < a.S) {
      const float t = t_table[j];
      px = ox + dx * t; py = oy + dy * t; pz = oz + dz * t;
      const float nrm = a.norm_l2 ? ug_norm3_torch(px, py, pz) : fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz)));
      if (!(nrm <= 1.0f)) {
        const float sc = a.B - (1.0f / nrm) * a.A;      // `A / norm` = reciprocal(norm) * A in torch (ug_contract)
        px = px / nrm * sc; py = py / nrm * sc; pz = pz / nrm * sc;
      }
      const float ux = ug_unorm(px, lox, hix), uy = ug_unorm(py, loy, hiy), uz = ug_unorm(pz, loz, hiz);
      auto level = [&](int l, float cx, float cy, float cz) -> float {
        const ug_taps tp = ug_tap_setup_ld(a.X, a.Y, a.Z, cx, cy, cz);
        const float *__restrict__ g = grid + (int64_t)l * vol;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += g[tp.off[c]] * tp.w[c];
        return acc;
      };
      // as k_grid_query: level 0, then the sin and the cos level of a frequency together (one sin / cos evaluation for both, 16 corner
      // loads in flight), sums added in level order: the densities stay bit-identical to the composed path's
      dens = level(0, ux, uy, uz);
      for (int k = 0; 2 * k + 2 < a.P; ++k) {
        const float f = (float)(1 << k);
        float sx, kx, sy, ky, sz, kz;
        ug_sincos(f * ux, &sx, &kx);
        ug_sincos(f * uy, &sy, &ky);
        ug_sincos(f * uz, &sz, &kz);
        const ug_taps ts = ug_tap_setup_ld(a.X, a.Y, a.Z, sx, sy, sz), tc = ug_tap_setup_ld(a.X, a.Y, a.Z, kx, ky, kz);
        const float *__restrict__ gs = grid + (int64_t)(2 * k + 1) * vol, *__restrict__ gc = grid + (int64_t)(2 * k + 2) * vol;
        float vs[8], vc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) vs[c] = gs[ts.off[c]];
#pragma unroll
        for (int c = 0; c < 8; ++c) vc[c] = gc[tc.off[c]];
        float as = 0.f, ac = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) as += vs[c] * ts.w[c];
#pragma unroll
        for (int c = 0; c < 8; ++c) ac += vc[c] * tc.w[c];
        dens = dens + as;
        dens = dens + ac;
      }
      if (a.F > 0) dens = dens / (float)a.P;
      float e;
      alpha = ug_train_alpha(dens, a.shift, a.interval, &e);
      keep = alpha > a.thres;
    }
    unsigned long long m = __ballot(keep);
    float myT = 1.f, myW = 0.f;
    if (W2) {
      unsigned long long mm = m;
      while (mm != 0ull) {
        const int k = __builtin_ctzll(mm);
        mm &= mm - 1ull;
        const float ak = ug_readlane_f(alpha, k);
        if (lane == k) {
          myT = T_cum;
          myW = T_cum * ak;
        }
        T_cum = (float)((double)T_cum * (1. - (double)ak));
        if ((double)T_cum < 1e-3) {          // the sample that crosses keeps its weight; the ray ends here
          stopped = true;
          m &= (2ull << k) - 1ull;
          keep = keep && lane <= k;
          break;
        }
      }
    }
    if (m != 0ull) {
      if (keep) {
        const int64_t idx = slot + kept + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        s_pts[3 * idx] = px; s_pts[3 * idx + 1] = py; s_pts[3 * idx + 2] = pz;
        s_dens[idx] = dens;
        s_step[idx] = j;
        if (W2) { s_w[idx] = myW; s_T[idx] = myT; }
      }
      kept += __popcll(m);
      if (W2) kept2 += __popcll(__ballot(keep && myW > a.thres));
    }
  }
  if (lane == 0) {
    count[ray] = kept;
    if (W2) { count2[ray] = kept2; alphainv_last[ray] = T_cum; }
  }
}

// k_train_march<true> for the two dense-grid models (ug_train_vox above): MODE 1 = DirectContractedVoxGO, 2 = DirectVoxGO.
// Same slot / scratch contract and the same stage-2 recurrence; P = 1, F = 0.
template <int MODE>
__global__ void __launch_bounds__(256)
k_train_march_vox(ug_train_args a, ug_train_vox v, const float *__restrict__ grid, const float *__restrict__ rays_o,
                  const float *__restrict__ rays_d, const float *__restrict__ t_table, const float *__restrict__ xyz_min,
                  const float *__restrict__ xyz_max, float *__restrict__ s_pts, float *__restrict__ s_dens,
                  int32_t *__restrict__ s_step, int32_t *__restrict__ count, float *__restrict__ s_w, float *__restrict__ s_T,
                  int32_t *__restrict__ count2, float *__restrict__ alphainv_last) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ray >= a.n_rays) return;
  const int lane = ug_lane();
  const float lox = xyz_min[0], loy = xyz_min[1], loz = xyz_min[2], hix = xyz_max[0], hiy = xyz_max[1], hiz = xyz_max[2];
  const float rox = rays_o[3 * ray], roy = rays_o[3 * ray + 1], roz = rays_o[3 * ray + 2];
  const float rdx = rays_d[3 * ray], rdy = rays_d[3 * ray + 1], rdz = rays_d[3 * ray + 2];
  float ox, oy, oz, dx, dy, dz;
  int n = a.S;               // samples of this ray (wave-uniform)
  if (MODE == 1) {
    ox = (rox - a.cx) / a.rx; oy = (roy - a.cy) / a.ry; oz = (roz - a.cz) / a.rz;
    const float dn = ug_norm3_torch(rdx, rdy, rdz);
    dx = rdx / dn; dy = rdy / dn; dz = rdz / dn;
  } else {
    // ray / box slab test; a zero direction component is replaced by float(1e-6) (infer_t_minmax)
    const float vx = (rdx == 0.f) ? (float)1e-6 : rdx, vy = (rdy == 0.f) ? (float)1e-6 : rdy, vz = (rdz == 0.f) ? (float)1e-6 : rdz;
    const float ax = (hix - rox) / vx, ay = (hiy - roy) / vy, az = (hiz - roz) / vz;
    const float bx = (lox - rox) / vx, by = (loy - roy) / vy, bz = (loz - roz) / vz;
    const float tmin = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), v.far), v.near);
    const float tmax = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), v.far), v.near);
    const float rn = sqrtf(rdx * rdx + rdy * rdy + rdz * rdz);
    const double c = (double)ceilf((tmax - tmin) * rn / v.stepdist);        // infer_n_samples
    const double nn = c > 1. ? c : 1.;
    n = nn > (double)a.S ? a.S : (int)nn;
    ox = rox + rdx * tmin; oy = roy + rdy * tmin; oz = roz + rdz * tmin;     // rays_start
    dx = rdx / rn; dy = rdy / rn; dz = rdz / rn;                             // rays_dir
  }
  const int64_t slot = ray * a.S;
  int kept = 0, kept2 = 0;   // wave-uniform
  float T_cum = 1.f;         // wave-uniform
  float cum = 0.f;           // wave-uniform: the cumdist_thres carry (MODE 1)
  bool stopped = false;
  for (int j0 = 0; j0 < n && !stopped; j0 += UG_WAVE) {
    const int j = j0 + lane;
    bool keep = false, inner = true;
    float px = 0.f, py = 0.f, pz = 0.f, dens = 0.f, alpha = 0.f;
    if (MODE == 1) {
      float dj = 0.f;
      if (j < n) {
        inner = ug_train_point(a, ox, oy, oz, dx, dy, dz, t_table[j], px, py, pz) <= 1.0f;
        if (j > 0) {         // |p_j - p_{j-1}| over ALL consecutive samples (dcvgo.py:287)
          float qx, qy, qz;
          ug_train_point(a, ox, oy, oz, dx, dy, dz, t_table[j - 1], qx, qy, qz);
          dj = ug_norm3_torch(px - qx, py - qy, pz - qz);
        }
      }
      bool over = false;
      const int cnt = (n - j0) < UG_WAVE ? (n - j0) : UG_WAVE;
      for (int k = (j0 == 0 ? 1 : 0); k < cnt; ++k) {       // mask[:, 1:] |= cumdist_thres(dist): sample 0 has no distance
        cum += ug_readlane_f(dj, k);
        const bool ov = cum > v.dist_thres;
        cum *= ov ? 0.f : 1.f;
        if (lane == k) over = ov;
      }
      keep = j < n && (inner || over);
    } else {
      if (j < n) {
        const float dist = v.stepdist * (float)j;
        px = ox + dx * dist; py = oy + dy * dist; pz = oz + dz * dist;
        keep = !((lox > px) | (loy > py) | (loz > pz) | (hix < px) | (hiy < py) | (hiz < pz));   // ~mask_outbbox
      }
    }
    if (keep) keep = ug_train_maskcache(v, px, py, pz);
    if (keep) {
      const float ux = ug_unorm(px, lox, hix), uy = ug_unorm(py, loy, hiy), uz = ug_unorm(pz, loz, hiz);
      const ug_taps tp = ug_tap_setup_ld(a.X, a.Y, a.Z, ux, uy, uz);
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc += grid[tp.off[c]] * tp.w[c];
      dens = acc;
      float e;
      alpha = ug_train_alpha(dens, a.shift, a.interval, &e);
      keep = alpha > a.thres;
    }
    unsigned long long m = __ballot(keep);
    float myT = 1.f, myW = 0.f;
    unsigned long long mm = m;
    while (mm != 0ull) {
      const int k = __builtin_ctzll(mm);
      mm &= mm - 1ull;
      const float ak = ug_readlane_f(alpha, k);
      if (lane == k) {
        myT = T_cum;
        myW = T_cum * ak;
      }
      T_cum = (float)((double)T_cum * (1. - (double)ak));
      if ((double)T_cum < 1e-3) {          // the sample that crosses keeps its weight; the ray ends here
        stopped = true;
        m &= (2ull << k) - 1ull;
        keep = keep && lane <= k;
        break;
      }
    }
    if (m != 0ull) {
      if (keep) {
        const int64_t idx = slot + kept + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        s_pts[3 * idx] = px; s_pts[3 * idx + 1] = py; s_pts[3 * idx + 2] = pz;
        s_dens[idx] = dens;
        s_step[idx] = (MODE == 1 && !inner) ? ~j : j;
        s_w[idx] = myW; s_T[idx] = myT;
      }
      kept += __popcll(m);
      kept2 += __popcll(__ballot(keep && myW > a.thres));
    }
  }
  if (lane == 0) {
    count[ray] = kept;
    count2[ray] = kept2;
    alphainv_last[ray] = T_cum;
  }
}

// after the two cumsums: one wave per ray copies its slot to the ray-major arrays of stage 1 (M1 samples: what the backward
// walks) and of stage 2 (M2 samples above the weight threshold: what the k0 lookup, the rgbnet and the loss consume);
// pos2[i] = the stage-2 index of stage-1 sample i or -1
__global__ void __launch_bounds__(256)
k_train_compact2(int64_t n_rays, int32_t S, float shift, float interval, float thres, const float *__restrict__ s_pts,
                 const float *__restrict__ s_dens, const int32_t *__restrict__ s_step, const float *__restrict__ s_w,
                 const float *__restrict__ s_T, const int32_t *__restrict__ count, const int64_t *__restrict__ end1,
                 const int32_t *__restrict__ count2, const int64_t *__restrict__ end2, const float *__restrict__ t_table,
                 float *__restrict__ pts1, float *__restrict__ dens1, float *__restrict__ w1, float *__restrict__ T1,
                 int32_t *__restrict__ pos2, float *__restrict__ pts2, float *__restrict__ dens2, float *__restrict__ alpha2,
                 float *__restrict__ w2, int64_t *__restrict__ ray_id2, int64_t *__restrict__ step_id2, float *__restrict__ tt2,
                 uint8_t *__restrict__ inner2, int64_t cap2) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ray >= n_rays) return;
  const int lane = ug_lane();
  const int n = count[ray];
  const int64_t d1 = end1[ray] - n, src = ray * (int64_t)S;
  int64_t d2 = end2[ray] - count2[ray];
  for (int i0 = 0; i0 < n; i0 += UG_WAVE) {
    const int i = i0 + lane;
    const bool on = i < n;
    float px = 0.f, py = 0.f, pz = 0.f, dn = 0.f, w = 0.f;
    int st = 0;
    bool inr = true;
    if (on) {
      px = s_pts[3 * (src + i)]; py = s_pts[3 * (src + i) + 1]; pz = s_pts[3 * (src + i) + 2];
      dn = s_dens[src + i]; w = s_w[src + i]; st = s_step[src + i];
      if (st < 0) { st = ~st; inr = false; }      // k_train_march_vox<1>: a contracted sample
      pts1[3 * (d1 + i)] = px; pts1[3 * (d1 + i) + 1] = py; pts1[3 * (d1 + i) + 2] = pz;
      dens1[d1 + i] = dn; w1[d1 + i] = w; T1[d1 + i] = s_T[src + i];
    }
    const bool k2 = on && w > thres;
    const unsigned long long m = __ballot(k2);
    const int64_t o = d2 + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    // cap2: rows the stage-2 arrays hold (sync-free step with a caller-chosen capacity: what exceeds it is dropped, memory-safe, and
    // the caller finds totals[1] > capacity when it next looks; everywhere else cap2 = INT64_MAX)
    if (on) pos2[d1 + i] = (k2 && o < cap2) ? (int32_t)o : -1;
    if (k2 && o < cap2) {
      float e;
      pts2[3 * o] = px; pts2[3 * o + 1] = py; pts2[3 * o + 2] = pz;
      dens2[o] = dn;
      alpha2[o] = ug_train_alpha(dn, shift, interval, &e);
      w2[o] = w;
      ray_id2[o] = ray;
      step_id2[o] = st;
      tt2[o] = t_table ? t_table[st] : (float)st;
      if (inner2) inner2[o] = inr ? 1 : 0;
    }
    d2 += __popcll(m);
  }
}

// backward of stage 2 of the sampling in one pass over the stage-1 samples of a ray, last to first: Alphas2Weights' backward
// (k_alpha2weight_bwd: float running sum from the last sample, per-sample double expression), Raw2Alpha's (k_raw2alpha_bwd:
// exp clamped at 1e10, double product), and the gradient that reaches the raw density directly (the loss's nearclip term on
// the stage-2 samples) added on top -- g_dens1[i] is what the density lookup's scatter consumes.
__global__ void __launch_bounds__(256)
k_train_sample_bwd(int64_t n_rays, float shift, float interval, const float *__restrict__ dens1, const float *__restrict__ w1,
                   const float *__restrict__ T1, const int32_t *__restrict__ pos2, const int32_t *__restrict__ count,
                   const int64_t *__restrict__ end1, const float *__restrict__ alphainv_last, const float *__restrict__ g_w2,
                   const float *__restrict__ g_last, const float *__restrict__ g_dens2, float *__restrict__ g_dens1) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= n_rays) return;
  const int lane = ug_lane();
  const int64_t i_e = end1[r], i_s = i_e - count[r];
  float back = g_last ? g_last[r] * alphainv_last[r] : 0.f;
  for (int64_t top = i_e; top > i_s; top -= UG_WAVE) {
    const int64_t i = top - 1 - lane;          // lane k holds sample top-1-k
    const bool ok = i >= i_s;
    const int p2 = ok ? pos2[i] : -1;
    const float gw = (p2 >= 0 && g_w2) ? g_w2[p2] : 0.f;
    const float prod = ok ? gw * w1[i] : 0.f;
    const int cnt = (int)((top - i_s) < UG_WAVE ? (top - i_s) : UG_WAVE);
    float my_back = 0.f;
    for (int k = 0; k < cnt; ++k) {
      if (lane == k) my_back = back;
      back += ug_readlane_f(prod, k);
    }
    if (ok) {
      float ef;
      const float a = ug_train_alpha(dens1[i], shift, interval, &ef);
      const float g_alpha = (float)((double)(gw * T1[i]) - (double)my_back / ((double)(1 - a) + 1e-10));
      const double e = (double)ef;
      const double em = e < 1e10 ? e : 1e10;
      const float pw = powf(1 + ef, -interval - 1);
      float g = (float)(em * (double)pw * (double)interval * (double)g_alpha);
      if (p2 >= 0 && g_dens2) g = g + g_dens2[p2];
      g_dens1[i] = g;
    }
  }
}

__global__ void __launch_bounds__(256)
k_train_compact(int64_t n_rays, int32_t S, const float *__restrict__ s_pts, const float *__restrict__ s_dens,
                const int32_t *__restrict__ s_step, const int32_t *__restrict__ count, const int64_t *__restrict__ offset_end,
                const float *__restrict__ t_table, float *__restrict__ pts, float *__restrict__ dens,
                int64_t *__restrict__ ray_id, int64_t *__restrict__ step_id, float *__restrict__ tt) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (ray >= n_rays) return;
  const int lane = ug_lane();
  const int n = count[ray];
  const int64_t dst = offset_end[ray] - n, src = ray * (int64_t)S;
  for (int i = lane; i < n; i += UG_WAVE) {
    const int st = s_step[src + i];
    pts[3 * (dst + i)] = s_pts[3 * (src + i)];
    pts[3 * (dst + i) + 1] = s_pts[3 * (src + i) + 1];
    pts[3 * (dst + i) + 2] = s_pts[3 * (src + i) + 2];
    dens[dst + i] = s_dens[src + i];
    ray_id[dst + i] = ray;
    step_id[dst + i] = st;
    tt[dst + i] = t_table[st];
  }
}

static int ug_train_march_any(const float *density_grid, int P, int X, int Y, int Z, int freq_num, const float *rays_o,
                              const float *rays_d, int64_t n_rays, const float *t_table, int32_t n_samples,
                              const float *scene_center3, const float *scene_radius3, const float *xyz_min,
                              const float *xyz_max, double bg_len, int norm_l2, float act_shift, float interval,
                              float thres, float *scratch_pts, float *scratch_density, int32_t *scratch_step,
                              int32_t *count, float *scratch_w, float *scratch_T, int32_t *count2, float *alphainv_last,
                              ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  if (P != (freq_num > 0 ? 2 * freq_num + 1 : 1) || n_samples <= 0) return (int)hipErrorInvalidValue;
  ug_train_args a;
  a.n_rays = n_rays; a.S = n_samples; a.P = P; a.F = freq_num; a.X = X; a.Y = Y; a.Z = Z; a.norm_l2 = norm_l2;
  a.cx = scene_center3[0]; a.cy = scene_center3[1]; a.cz = scene_center3[2];
  a.rx = scene_radius3[0]; a.ry = scene_radius3[1]; a.rz = scene_radius3[2];
  const double Bd = 1.0 + bg_len;
  a.B = (float)Bd; a.A = (float)(Bd * 1.0 - 1.0);
  a.shift = act_shift; a.interval = interval; a.thres = thres;
  if (scratch_w)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_train_march<true>), dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), a, density_grid,
                       rays_o, rays_d, t_table, xyz_min, xyz_max, scratch_pts, scratch_density, scratch_step, count, scratch_w, scratch_T,
                       count2, alphainv_last);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_train_march<false>), dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), a, density_grid,
                       rays_o, rays_d, t_table, xyz_min, xyz_max, scratch_pts, scratch_density, scratch_step, count, (float *)nullptr,
                       (float *)nullptr, (int32_t *)nullptr, (float *)nullptr);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_train_march(const float *density_grid, int P, int X, int Y, int Z, int freq_num, const float *rays_o,
                                 const float *rays_d, int64_t n_rays, const float *t_table, int32_t n_samples,
                                 const float *scene_center3, const float *scene_radius3, const float *xyz_min,
                                 const float *xyz_max, double bg_len, int norm_l2, float act_shift, float interval,
                                 float thres, float *scratch_pts, float *scratch_density, int32_t *scratch_step,
                                 int32_t *count, ugrid_stream_t s) {
  return ug_train_march_any(density_grid, P, X, Y, Z, freq_num, rays_o, rays_d, n_rays, t_table, n_samples, scene_center3, scene_radius3,
                            xyz_min, xyz_max, bg_len, norm_l2, act_shift, interval, thres, scratch_pts, scratch_density, scratch_step,
                            count, nullptr, nullptr, nullptr, nullptr, s);
}

extern "C" int ugrid_train_sample(const float *density_grid, int P, int X, int Y, int Z, int freq_num, const float *rays_o,
                                  const float *rays_d, int64_t n_rays, const float *t_table, int32_t n_samples,
                                  const float *scene_center3, const float *scene_radius3, const float *xyz_min,
                                  const float *xyz_max, double bg_len, int norm_l2, float act_shift, float interval,
                                  float thres, float *scratch_pts, float *scratch_density, int32_t *scratch_step,
                                  float *scratch_w, float *scratch_T, int32_t *count, int32_t *count2, float *alphainv_last,
                                  ugrid_stream_t s) {
  if (!scratch_w || !scratch_T || !count2 || !alphainv_last) return (int)hipErrorInvalidValue;
  return ug_train_march_any(density_grid, P, X, Y, Z, freq_num, rays_o, rays_d, n_rays, t_table, n_samples, scene_center3, scene_radius3,
                            xyz_min, xyz_max, bg_len, norm_l2, act_shift, interval, thres, scratch_pts, scratch_density, scratch_step,
                            count, scratch_w, scratch_T, count2, alphainv_last, s);
}

extern "C" int ugrid_train_sample_compact(int64_t n_rays, int32_t n_samples, float act_shift, float interval, float thres,
                                          const float *scratch_pts, const float *scratch_density, const int32_t *scratch_step,
                                          const float *scratch_w, const float *scratch_T, const int32_t *count,
                                          const int64_t *offset_end, const int32_t *count2, const int64_t *offset_end2,
                                          const float *t_table, float *pts1, float *density1, float *weights1, float *T1,
                                          int32_t *pos2, float *pts2, float *density2, float *alpha2, float *weights2,
                                          int64_t *ray_id2, int64_t *step_id2, float *t2, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(k_train_compact2, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), n_rays, n_samples, act_shift, interval,
                     thres, scratch_pts, scratch_density, scratch_step, scratch_w, scratch_T, count, offset_end, count2, offset_end2,
                     t_table, pts1, density1, weights1, T1, pos2, pts2, density2, alpha2, weights2, ray_id2, step_id2, t2,
                     (uint8_t *)nullptr, ug_tl_devn.cap > 0 ? ug_tl_devn.cap : INT64_MAX);
  UG_LAUNCH_CHECK();
  return 0;
}

static int ug_fill_train_vox(ug_train_vox *v, const uint8_t *mask, const int32_t *mask_dims3, const float *xyz2ijk_scale3,
                             const float *xyz2ijk_shift3) {
  if (!mask || !mask_dims3 || !xyz2ijk_scale3 || !xyz2ijk_shift3) return (int)hipErrorInvalidValue;
  v->mask = mask;
  v->mi = mask_dims3[0]; v->mj = mask_dims3[1]; v->mk = mask_dims3[2];
  v->sx = xyz2ijk_scale3[0]; v->sy = xyz2ijk_scale3[1]; v->sz = xyz2ijk_scale3[2];
  v->hx = xyz2ijk_shift3[0]; v->hy = xyz2ijk_shift3[1]; v->hz = xyz2ijk_shift3[2];
  v->dist_thres = 0.f; v->near = 0.f; v->far = 0.f; v->stepdist = 1.f;
  return 0;
}

extern "C" int ugrid_train_sample_dcvgo(const float *density_grid, int X, int Y, int Z, const float *rays_o, const float *rays_d,
                                        int64_t n_rays, const float *t_table, int32_t n_samples, const float *scene_center3,
                                        const float *scene_radius3, const float *xyz_min, const float *xyz_max, double bg_len,
                                        int norm_l2, float dist_thres, const uint8_t *mask, const int32_t *mask_dims3,
                                        const float *xyz2ijk_scale3, const float *xyz2ijk_shift3, float act_shift, float interval,
                                        float thres, float *scratch_pts, float *scratch_density, int32_t *scratch_step,
                                        float *scratch_w, float *scratch_T, int32_t *count, int32_t *count2, float *alphainv_last,
                                        ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  if (n_samples <= 0 || !scratch_w || !scratch_T || !count2 || !alphainv_last || !t_table) return (int)hipErrorInvalidValue;
  ug_train_vox v;
  const int rc = ug_fill_train_vox(&v, mask, mask_dims3, xyz2ijk_scale3, xyz2ijk_shift3);
  if (rc) return rc;
  v.dist_thres = dist_thres;
  ug_train_args a;
  a.n_rays = n_rays; a.S = n_samples; a.P = 1; a.F = 0; a.X = X; a.Y = Y; a.Z = Z; a.norm_l2 = norm_l2;
  a.cx = scene_center3[0]; a.cy = scene_center3[1]; a.cz = scene_center3[2];
  a.rx = scene_radius3[0]; a.ry = scene_radius3[1]; a.rz = scene_radius3[2];
  const double Bd = 1.0 + bg_len;
  a.B = (float)Bd; a.A = (float)bg_len;       // dcvgo.py:261: (1 + bg_len) - bg_len / norm
  a.shift = act_shift; a.interval = interval; a.thres = thres;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_train_march_vox<1>), dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), a, v, density_grid,
                     rays_o, rays_d, t_table, xyz_min, xyz_max, scratch_pts, scratch_density, scratch_step, count, scratch_w, scratch_T,
                     count2, alphainv_last);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_train_sample_dvgo(const float *density_grid, int X, int Y, int Z, const float *rays_o, const float *rays_d,
                                       int64_t n_rays, int32_t slots_per_ray, const float *xyz_min, const float *xyz_max, float near,
                                       float far, float stepdist, const uint8_t *mask, const int32_t *mask_dims3,
                                       const float *xyz2ijk_scale3, const float *xyz2ijk_shift3, float act_shift, float interval,
                                       float thres, float *scratch_pts, float *scratch_density, int32_t *scratch_step,
                                       float *scratch_w, float *scratch_T, int32_t *count, int32_t *count2, float *alphainv_last,
                                       ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  if (slots_per_ray <= 0 || !(stepdist > 0.f) || !scratch_w || !scratch_T || !count2 || !alphainv_last) return (int)hipErrorInvalidValue;
  ug_train_vox v;
  const int rc = ug_fill_train_vox(&v, mask, mask_dims3, xyz2ijk_scale3, xyz2ijk_shift3);
  if (rc) return rc;
  v.near = near; v.far = far; v.stepdist = stepdist;
  ug_train_args a;
  a.n_rays = n_rays; a.S = slots_per_ray; a.P = 1; a.F = 0; a.X = X; a.Y = Y; a.Z = Z; a.norm_l2 = 0;
  a.cx = a.cy = a.cz = 0.f; a.rx = a.ry = a.rz = 1.f; a.B = 1.f; a.A = 0.f;
  a.shift = act_shift; a.interval = interval; a.thres = thres;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_train_march_vox<2>), dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), a, v, density_grid,
                     rays_o, rays_d, (const float *)nullptr, xyz_min, xyz_max, scratch_pts, scratch_density, scratch_step, count, scratch_w,
                     scratch_T, count2, alphainv_last);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_train_sample_compact_vox(int64_t n_rays, int32_t slots_per_ray, float act_shift, float interval, float thres,
                                              const float *scratch_pts, const float *scratch_density, const int32_t *scratch_step,
                                              const float *scratch_w, const float *scratch_T, const int32_t *count,
                                              const int64_t *offset_end, const int32_t *count2, const int64_t *offset_end2,
                                              const float *t_table, float *pts1, float *density1, float *weights1, float *T1,
                                              int32_t *pos2, float *pts2, float *density2, float *alpha2, float *weights2,
                                              int64_t *ray_id2, int64_t *step_id2, float *t2, uint8_t *inner2, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(k_train_compact2, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), n_rays, slots_per_ray, act_shift,
                     interval, thres, scratch_pts, scratch_density, scratch_step, scratch_w, scratch_T, count, offset_end, count2,
                     offset_end2, t_table, pts1, density1, weights1, T1, pos2, pts2, density2, alpha2, weights2, ray_id2, step_id2, t2,
                     inner2, ug_tl_devn.cap > 0 ? ug_tl_devn.cap : INT64_MAX);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_train_sample_backward(int64_t n_rays, float act_shift, float interval, const float *density1,
                                           const float *weights1, const float *T1, const int32_t *pos2, const int32_t *count,
                                           const int64_t *offset_end, const float *alphainv_last, const float *g_weights2,
                                           const float *g_alphainv_last, const float *g_density2, float *g_density1,
                                           ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(k_train_sample_bwd, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), n_rays, act_shift, interval,
                     density1, weights1, T1, pos2, count, offset_end, alphainv_last, g_weights2, g_alphainv_last, g_density2, g_density1);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_train_compact(int64_t n_rays, int32_t n_samples, const float *scratch_pts, const float *scratch_density,
                                   const int32_t *scratch_step, const int32_t *count, const int64_t *offset_end,
                                   const float *t_table, float *pts, float *density, int64_t *ray_id, int64_t *step_id,
                                   float *t, ugrid_stream_t s) {
  if (n_rays <= 0) return 0;
  hipLaunchKernelGGL(k_train_compact, dim3(ug_blocks(n_rays * UG_WAVE, 256)), dim3(256), 0, ST(s), n_rays, n_samples,
                     scratch_pts, scratch_density, scratch_step, count, offset_end, t_table, pts, density, ray_id, step_id, t);
  UG_LAUNCH_CHECK();
  return 0;
}

template <int F, bool L2, int W>
__global__ void __launch_bounds__(256, W)
k_march(ug_march_args a, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
        const float *__restrict__ t_table, const float *__restrict__ s_table,
        const float *__restrict__ bricks, float *__restrict__ alphainv_last, float *__restrict__ depth,
        ug_ws_view ws, int64_t nblocks) {
  const int64_t blk = ug_xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t tile = blk * 4 + (threadIdx.x >> 6);
  if (tile >= ws.n_tiles) return;
  const int n = ug_march_tile<F, L2>(a, rays_o, rays_d, t_table, s_table, bricks, alphainv_last, depth, tile,
                                     ws.ent + tile * ws.cap, ws.slot + tile * ws.cap);
  if (ug_lane() == 0) ws.count[tile] = n;
}


// DirectContractedVoxGO march (single-level grids: F = 0): k_march + the cumdist_thres rule + the mask cache + wsum_mid
template <bool L2>
__global__ void __launch_bounds__(256, 5)
k_march_dcvgo(ug_march_args a, ug_dc_args dc, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
              const float *__restrict__ t_table, const float *__restrict__ s_table, const float *__restrict__ bricks,
              float *__restrict__ alphainv_last, float *__restrict__ depth, float *__restrict__ wsum_mid, ug_ws_view ws,
              int64_t nblocks) {
  const int64_t blk = ug_xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t tile = blk * 4 + (threadIdx.x >> 6);
  if (tile >= ws.n_tiles) return;
  const int n = ug_march_tile<0, L2, true>(a, rays_o, rays_d, t_table, s_table, bricks, alphainv_last, depth, tile,
                                           ws.ent + tile * ws.cap, ws.slot + tile * ws.cap, dc, wsum_mid);
  if (ug_lane() == 0) ws.count[tile] = n;
}

// bounded DirectVoxGO march (ug_march_tile_dvgo): variable-length rays clipped against the scene box
__global__ void __launch_bounds__(256, 6)
k_march_dvgo(ug_march_args a, ug_dv_args dv, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
             const float *__restrict__ bricks, float *__restrict__ alphainv_last, float *__restrict__ depth, ug_ws_view ws,
             int64_t nblocks) {
  const int64_t blk = ug_xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t tile = blk * 4 + (threadIdx.x >> 6);
  if (tile >= ws.n_tiles) return;
  const int n = ug_march_tile_dvgo(a, dv, rays_o, rays_d, bricks, alphainv_last, depth, tile, ws.ent + tile * ws.cap,
                                   ws.slot + tile * ws.cap, a.S);
  if (ug_lane() == 0) ws.count[tile] = n;
}

// ----------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------
extern "C" int64_t ugrid_render_ws_bytes(int64_t n_rays, int32_t S) {
  const int64_t n_tiles = (n_rays + UG_WAVE - 1) / UG_WAVE, cap = (int64_t)UG_WAVE * S;
  return 256 + ug_align256(n_tiles * 4) + ug_align256(n_tiles * cap * 16) + ug_align256(n_tiles * cap);
}

static int ug_grid_query_any(bool cl, const float *grid, int P, int C, int X, int Y, int Z, const float *xyz,
                             const float *xyz_min, const float *xyz_max, int freq_num, int64_t n, float *out, hipStream_t st) {
  if (n <= 0) return 0;
  if (P != (freq_num > 0 ? 2 * freq_num + 1 : 1)) return (int)hipErrorInvalidValue;
  const int64_t nl = ug_launch_rows(n);          // (a device count in force: the grid follows the hint, the kernel the count)
  if (cl)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_query<true>), dim3(ug_blocks(nl * C, 256)), dim3(256), 0, st, grid, P, C, X, Y, Z, xyz,
                       xyz_min, xyz_max, freq_num, n, out, ug_tl_devn.ptr);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_query<false>), dim3(ug_blocks(nl, 256)), dim3(256), 0, st, grid, P, C, X, Y, Z, xyz,
                       xyz_min, xyz_max, freq_num, n, out, ug_tl_devn.ptr);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_grid_query(const float *grid, int P, int C, int X, int Y, int Z, const float *xyz,
                                const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                float *out, ugrid_stream_t s) {
  return ug_grid_query_any(false, grid, P, C, X, Y, Z, xyz, xyz_min, xyz_max, freq_num, n, out, ST(s));
}

extern "C" int ugrid_grid_query_cl(const float *grid, int P, int C, int X, int Y, int Z, const float *xyz,
                                   const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                   float *out, ugrid_stream_t s) {
  return ug_grid_query_any(true, grid, P, C, X, Y, Z, xyz, xyz_min, xyz_max, freq_num, n, out, ST(s));
}

static int ug_grid_query_backward_any(bool cl, const float *grad_out, int P, int C, int X, int Y, int Z, const float *xyz,
                                      const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                      float *grad_grid, hipStream_t st, uint32_t *touch = nullptr) {
  if (n <= 0) return 0;
  if (P != (freq_num > 0 ? 2 * freq_num + 1 : 1)) return (int)hipErrorInvalidValue;
  const int64_t nl = ug_launch_rows(n);
  if (cl)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_query_backward<true>), dim3(ug_blocks(nl * P * C, 256)), dim3(256), 0, st, grad_out, P, C,
                       X, Y, Z, xyz, xyz_min, xyz_max, freq_num, n, grad_grid, touch, ug_tl_devn.ptr);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_grid_query_backward<false>), dim3(ug_blocks(nl * P, 256)), dim3(256), 0, st, grad_out, P, C,
                       X, Y, Z, xyz, xyz_min, xyz_max, freq_num, n, grad_grid, (uint32_t *)nullptr, ug_tl_devn.ptr);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_grid_query_backward(const float *grad_out, int P, int C, int X, int Y, int Z, const float *xyz,
                                         const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                         float *grad_grid, ugrid_stream_t s) {
  return ug_grid_query_backward_any(false, grad_out, P, C, X, Y, Z, xyz, xyz_min, xyz_max, freq_num, n, grad_grid, ST(s));
}

extern "C" int ugrid_grid_query_backward_cl(const float *grad_out, int P, int C, int X, int Y, int Z, const float *xyz,
                                            const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                            float *grad_grid, ugrid_stream_t s) {
  return ug_grid_query_backward_any(true, grad_out, P, C, X, Y, Z, xyz, xyz_min, xyz_max, freq_num, n, grad_grid, ST(s));
}

extern "C" int ugrid_grid_query_backward_cl_touch(const float *grad_out, int P, int C, int X, int Y, int Z, const float *xyz,
                                                  const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                                  float *grad_grid, uint32_t *touch, ugrid_stream_t s) {
  return ug_grid_query_backward_any(true, grad_out, P, C, X, Y, Z, xyz, xyz_min, xyz_max, freq_num, n, grad_grid, ST(s), touch);
}

static inline int ug_brick_ch(int C, int *H) {
  // density (C==1): 1 half x 1 channel; rgbnet-less k0 (C==3 handled by caller via halves=1, CH=4);
  // feature grids: 2 halves x ceil(C/2)
  if (C == 1) { *H = 1; return 1; }
  *H = 2;
  return UG_CH(C);
}

extern "C" int64_t ugrid_brick_bytes(int P, int C, int X, int Y, int Z, int direct) {
  int H, CH = ug_brick_ch(C, &H);
  if (direct) { H = 1; CH = 4; }
  return (int64_t)P * (X - 1) * (Y - 1) * (Z - 1) * H * 8 * CH * (int64_t)sizeof(float);
}

extern "C" int ugrid_pack_bricks(const float *grid, int P, int C, int X, int Y, int Z, int direct,
                                 float *bricks, ugrid_stream_t s) {
  if (X < 2 || Y < 2 || Z < 2 || P < 1 || C < 1) return (int)hipErrorInvalidValue;
  int H, CH = ug_brick_ch(C, &H);
  if (direct) { H = 1; CH = 4; }
  const int64_t total = (int64_t)P * (X - 1) * (Y - 1) * (Z - 1) * H * 8 * CH;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (C == 12 && !direct) {   // 384 B per cell in the quad layout (same size as two 192-byte halves)
    // multi-level grids are addressed with 32-bit byte offsets inside a level; a single-level grid (P == 1) with 64-bit ones
    if (P > 1 && (int64_t)(X - 1) * (Y - 1) * (Z - 1) * 384 >= ((int64_t)1 << 32)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pack_quad, dim3((unsigned)blocks), dim3(256), 0, ST(s), grid, P, C, X, Y, Z, bricks, total);
    UG_LAUNCH_CHECK();
    return 0;
  }
  int coef = direct ? 0 : 1;
  hipLaunchKernelGGL(k_pack_bricks, dim3((unsigned)blocks), dim3(256), 0, ST(s), grid, P, C, X, Y, Z, H, CH,
                     coef, bricks, total);
  UG_LAUNCH_CHECK();
  return 0;
}

static int g_march_waves = 6;
extern "C" int ug_set_march_waves(int w) { if (w < 4 || w > 6) return 1; g_march_waves = w; return 0; }
  // min waves/SIMD the march kernel is compiled for (register cap 512/W)

template <int F, int W>
static int ug_march_launch_w(const ugrid_render_params *p, const ug_march_args &a, const float *rays_o,
                             const float *rays_d, const float *t_table, const float *s_table,
                             const float *bricks, float *alphainv_last, float *depth, ug_ws_view ws,
                             hipStream_t st) {
  const int64_t nblocks = (ws.n_tiles + 3) / 4;
  const int64_t grid = ((nblocks + 7) / 8) * 8;  // room for the XCD remap
  if (p->norm_l2)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_march<F, true, W>), dim3((unsigned)grid), dim3(256), 0, st, a, rays_o,
                       rays_d, t_table, s_table, bricks, alphainv_last, depth, ws, nblocks);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_march<F, false, W>), dim3((unsigned)grid), dim3(256), 0, st, a, rays_o,
                       rays_d, t_table, s_table, bricks, alphainv_last, depth, ws, nblocks);
  UG_LAUNCH_CHECK();
  return 0;
}

template <int F>
static int ug_march_launch(const ugrid_render_params *p, const ug_march_args &a, const float *rays_o,
                           const float *rays_d, const float *t_table, const float *s_table,
                           const float *bricks, float *alphainv_last, float *depth, ug_ws_view ws,
                           hipStream_t st) {
  switch (g_march_waves) {
    case 4: return ug_march_launch_w<F, 4>(p, a, rays_o, rays_d, t_table, s_table, bricks, alphainv_last, depth, ws, st);
    case 6: return ug_march_launch_w<F, 6>(p, a, rays_o, rays_d, t_table, s_table, bricks, alphainv_last, depth, ws, st);
    default: return ug_march_launch_w<F, 5>(p, a, rays_o, rays_d, t_table, s_table, bricks, alphainv_last, depth, ws, st);
  }
}

extern "C" int ugrid_render_march(const ugrid_render_params *p, const float *rays_o, const float *rays_d,
                                  const float *t_table, const float *s_table, const float *density_bricks,
                                  float *alphainv_last, float *depth, void *ws_mem, ugrid_stream_t s) {
  if (p->n_rays <= 0) return 0;
  ug_march_args a;
  const int rc = ug_fill_march_args(p, a);
  if (rc) return rc;
  ug_ws_view ws = ug_ws_make(ws_mem, p->n_rays, p->n_samples);
  switch (p->freq_num) {
    case 1: return ug_march_launch<1>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    case 2: return ug_march_launch<2>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    case 3: return ug_march_launch<3>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    case 4: return ug_march_launch<4>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    case 5: return ug_march_launch<5>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    default: return (int)hipErrorInvalidValue;
  }
}


extern "C" int ugrid_render_march_dcvgo(const ugrid_render_params *p, const ugrid_dcvgo_params *q, const float *rays_o,
                                        const float *rays_d, const float *t_table, const float *s_table,
                                        const float *density_bricks, float *alphainv_last, float *depth, float *wsum_mid,
                                        void *ws_mem, ugrid_stream_t s) {
  if (p->n_rays <= 0) return 0;
  if (p->freq_num != 0 || !q || !q->mask || q->mask_x < 1 || q->mask_y < 1 || q->mask_z < 1) return (int)hipErrorInvalidValue;
  ug_march_args a;
  const int rc = ug_fill_march_args(p, a);
  if (rc) return rc;
  ug_dc_args dc;
  dc.mask = q->mask; dc.mi = q->mask_x; dc.mj = q->mask_y; dc.mk = q->mask_z;
  dc.sx = q->xyz2ijk_scale[0]; dc.sy = q->xyz2ijk_scale[1]; dc.sz = q->xyz2ijk_scale[2];
  dc.hx = q->xyz2ijk_shift[0]; dc.hy = q->xyz2ijk_shift[1]; dc.hz = q->xyz2ijk_shift[2];
  dc.dist_thres = q->dist_thres;
  ug_ws_view ws = ug_ws_make(ws_mem, p->n_rays, p->n_samples);
  const int64_t nblocks = (ws.n_tiles + 3) / 4;
  const int64_t grid = ((nblocks + 7) / 8) * 8;  // room for the XCD remap
  if (p->norm_l2)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_march_dcvgo<true>), dim3((unsigned)grid), dim3(256), 0, ST(s), a, dc, rays_o, rays_d, t_table,
                       s_table, density_bricks, alphainv_last, depth, wsum_mid, ws, nblocks);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_march_dcvgo<false>), dim3((unsigned)grid), dim3(256), 0, ST(s), a, dc, rays_o, rays_d, t_table,
                       s_table, density_bricks, alphainv_last, depth, wsum_mid, ws, nblocks);
  UG_LAUNCH_CHECK();
  return 0;
}


extern "C" int ugrid_render_march_dvgo(const ugrid_render_params *p, const ugrid_dvgo_params *q, const float *rays_o,
                                       const float *rays_d, const float *density_bricks, float *alphainv_last, float *depth,
                                       void *ws_mem, ugrid_stream_t s) {
  if (p->n_rays <= 0) return 0;
  if (p->freq_num != 0 || !q || !q->mask || q->mask_x < 1 || q->mask_y < 1 || q->mask_z < 1 || !(q->stepdist > 0.f))
    return (int)hipErrorInvalidValue;
  ug_march_args a;
  const int rc = ug_fill_march_args(p, a);
  if (rc) return rc;
  ug_dv_args dv;
  dv.mask = q->mask; dv.mi = q->mask_x; dv.mj = q->mask_y; dv.mk = q->mask_z;
  dv.sx = q->xyz2ijk_scale[0]; dv.sy = q->xyz2ijk_scale[1]; dv.sz = q->xyz2ijk_scale[2];
  dv.hx = q->xyz2ijk_shift[0]; dv.hy = q->xyz2ijk_shift[1]; dv.hz = q->xyz2ijk_shift[2];
  dv.near = q->near_clip; dv.far = q->far_clip; dv.stepdist = q->stepdist;
  ug_ws_view ws = ug_ws_make(ws_mem, p->n_rays, p->n_samples);
  const int64_t nblocks = (ws.n_tiles + 3) / 4;
  const int64_t grid = ((nblocks + 7) / 8) * 8;  // room for the XCD remap
  hipLaunchKernelGGL(k_march_dvgo, dim3((unsigned)grid), dim3(256), 0, ST(s), a, dv, rays_o, rays_d, density_bricks, alphainv_last,
                     depth, ws, nblocks);
  UG_LAUNCH_CHECK();
  return 0;
}


