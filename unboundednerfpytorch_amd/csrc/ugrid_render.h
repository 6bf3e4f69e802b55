// ugrid_render.h -- device code of the fused FourierGrid render path for gfx950 (MI355X), shared by
// ugrid_march.hip and ugrid_shade.hip (both compiled with -fno-slp-vectorize: packed v_pk_*_f32 math issues at
// half rate on gfx950 and costs extra moves -- a 16 % loss for the VALU-bound march kernel, 1.3 % for shade).
//
// Replaces, for inference, the torch-op chain of the reference's FourierGridModel.forward
// (FourierGrid/FourierGrid_model.py:509-672) and FourierGrid.forward (FourierGrid_grid.py:60-78):
//
//   k_march : 1 lane = 1 ray, 1 wave = 64 consecutive rays.  Per sample: contraction, Fourier
//             level coordinates, ONE 32-byte brick load per level (the 8 coefficients of the cell's
//             trilinear polynomial), 7 FMAs, mean over levels, raw2alpha, the two
//             thresholds and the front-to-back transmittance recurrence -- which is a plain serial
//             multiply in the lane's registers because a lane owns a ray.  A wave leaves the sample
//             loop as soon as all of its 64 rays have terminated (T < 1e-3) -- wave-level early
//             termination by ballot.  Surviving samples are compacted per wave (ballot + mbcnt
//             prefix) into that wave's private slice of the work list: no atomics, deterministic.
//   k_shade : 1 wave walks one tile's survivor list 32 at a time.  Lanes l and l+32 form a pair
//             that owns survivor (l&31): each gathers half of the k0 channels from the k0 bricks and
//             half of the view-direction embedding, which makes their registers exactly the B operand
//             of the MFMA (B[k-half = l>>5][j = l&31]).  The rgbnet runs "transposed" (H^T = W . X^T) so
//             every layer's accumulator registers are directly the next layer's B operands --
//             activations never leave the register file; packed weights (A operands) are read from
//             LDS.  Three fp32-accurate arithmetic modes: fp32 MFMA (bit-wise an fmaf chain), bf16x3
//             (three-way bf16 split, 6 products) and fp16x2 (power-of-two scaled two-way fp16 split,
//             3 products; the default when ugrid_pack_mlp finds the value ranges fit).
//
// Compiled with -ffp-contract=off: every a*b+c below that must match torch's separate
// multiply/add is written as such; fmaf is explicit where a fused operation is intended.
#pragma once
#include "ugrid_common.h"
#include "ugrid_math.h"
#include <string.h>

#define UG_CH(C) (2 * (((C) + 3) / 4))  // channels per half-brick (even: stored as channel pairs)
#define UG_MAX_F 5  // fourier_freq_num <= 5  (P <= 11 levels)


typedef float f32x16 __attribute__((ext_vector_type(16)));

// ----------------------------------------------------------------------------------------------
// shared per-sample math
// ----------------------------------------------------------------------------------------------
struct ug_vec3 { float x, y, z; };

// torch.linalg.vector_norm over 3 components (CPU kernel = fma chain, verified in tests)
__device__ __forceinline__ float ug_norm3_torch(float x, float y, float z) {
  return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}

// FourierGrid_model.py:534-548: p/|p| * ((1+bg) - bg/|p|) outside the unit cube (inf) / ball (l2).  `A / norm` with a Python
// number on the left is torch's Tensor.__rtruediv__ = reciprocal(norm) * A -- two roundings, formed the same way here
template <bool L2>
__device__ __forceinline__ ug_vec3 ug_contract(ug_vec3 p, float B, float A) {
  const float nrm = L2 ? ug_norm3_torch(p.x, p.y, p.z) : fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z)));
  if (!(nrm <= 1.0f)) {
    const float sc = B - (1.0f / nrm) * A;
    p.x = p.x / nrm * sc;
    p.y = p.y / nrm * sc;
    p.z = p.z / nrm * sc;
  }
  return p;
}

// grid_sample(align_corners=True) cell + fractional position along one axis of size n (n >= 2).
// Returns the cell index clamped to [0, n-2]; lo/hi are the two linear weights (x1 - ix), (ix - x0)
// exactly as torch forms them.  Points outside [-1,1] get zero weight on out-of-range corners
// (zero padding) -- `ok_lo/ok_hi` report whether each corner is inside the grid.
struct ug_axis { int cell; float wlo, whi; };

__device__ __forceinline__ ug_axis ug_axis_setup(float c, int n) {
  const float ix = ((c + 1.f) / 2.f) * (float)(n - 1);
  const float f0 = floorf(ix);
  ug_axis a;
  float wlo = (f0 + 1.f) - ix;  // weight of corner x0
  float whi = ix - f0;          // weight of corner x0+1
  // zero padding: a corner outside [0, n-1] contributes nothing
  int i0 = (int)fminf(fmaxf(f0, -2.f), (float)n);  // also tames NaN/inf
  if (i0 < 0 || i0 > n - 1) wlo = 0.f;
  if (i0 + 1 < 0 || i0 + 1 > n - 1) whi = 0.f;
  // re-express on a cell inside [0, n-2] so that one brick covers both corners
  if (i0 < 0) {            // only corner x0+1 (== 0) can be live: it is the LOW corner of cell 0
    a.cell = 0; a.wlo = (i0 == -1) ? whi : 0.f; a.whi = 0.f;
  } else if (i0 > n - 2) { // only corner x0 (== n-1) can be live: it is the HIGH corner of cell n-2
    a.cell = n - 2; a.whi = (i0 == n - 1) ? wlo : 0.f; a.wlo = 0.f;
  } else {
    a.cell = i0; a.wlo = wlo; a.whi = whi;
  }
  return a;
}

// the 8 trilinear weights in grid_sample's accumulation order: corner c = di*4 + dj*2 + dk, with
// i (world x, grid dim X) slowest.  torch forms each as (wx * wy) * wz with x = the W axis = world z.
struct ug_cellw { int64_t rec; float w[8]; };

__device__ __forceinline__ ug_cellw ug_cell_setup(float ux, float uy, float uz, int X, int Y, int Z,
                                                 int64_t level_base) {
  const ug_axis ax = ug_axis_setup(ux, X), ay = ug_axis_setup(uy, Y), az = ug_axis_setup(uz, Z);
  ug_cellw r;
  r.rec = level_base + ((int64_t)ax.cell * (Y - 1) + ay.cell) * (Z - 1) + az.cell;
  // torch: tnw = (ix_bse-ix)*(iy_bse-iy)*(iz_bse-iz) with its x = world z, y = world y, z = world x
  r.w[0] = az.wlo * ay.wlo * ax.wlo;
  r.w[1] = az.whi * ay.wlo * ax.wlo;
  r.w[2] = az.wlo * ay.whi * ax.wlo;
  r.w[3] = az.whi * ay.whi * ax.wlo;
  r.w[4] = az.wlo * ay.wlo * ax.whi;
  r.w[5] = az.whi * ay.wlo * ax.whi;
  r.w[6] = az.wlo * ay.whi * ax.whi;
  r.w[7] = az.whi * ay.whi * ax.whi;
  return r;
}

// level coordinate ℓ of the Fourier embedding of u: ℓ=0: u; ℓ=2k+1: sin(2^k u); ℓ=2k+2: cos(2^k u)
template <int F>
struct ug_levels { float cx[2 * F + 1], cy[2 * F + 1], cz[2 * F + 1]; };

template <int F>
__device__ __forceinline__ ug_levels<F> ug_pe(float ux, float uy, float uz) {
  ug_levels<F> L;
  L.cx[0] = ux; L.cy[0] = uy; L.cz[0] = uz;
#pragma unroll
  for (int k = 0; k < F; ++k) {
    const float f = (float)(1 << k);
    float s, c;
    sincosf(f * ux, &s, &c); L.cx[2 * k + 1] = s; L.cx[2 * k + 2] = c;
    sincosf(f * uy, &s, &c); L.cy[2 * k + 1] = s; L.cy[2 * k + 2] = c;
    sincosf(f * uz, &s, &c); L.cz[2 * k + 1] = s; L.cz[2 * k + 2] = c;
  }
  return L;
}

// world position -> normalised grid coordinate per axis: ((p - min) / (max - min)) * 2 - 1
__device__ __forceinline__ float ug_unorm(float p, float lo, float hi) {
  return ((p - lo) / (hi - lo)) * 2.f - 1.f;
}

// ----------------------------------------------------------------------------------------------
// work list shared by march and shade
// ----------------------------------------------------------------------------------------------
struct ug_ws_view {
  int32_t *count;   // [n_tiles]
  float4 *ent;      // [n_tiles][64*S]  (px, py, pz, weight)
  uint8_t *slot;    // [n_tiles][64*S]  ray slot (0..63) inside the tile
  int64_t n_tiles, cap;
};
#define UG_FEAT_STRIDE 12

__host__ __device__ static inline int64_t ug_align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

static inline ug_ws_view ug_ws_make(void *ws, int64_t n_rays, int32_t S) {
  ug_ws_view v;
  v.n_tiles = (n_rays + UG_WAVE - 1) / UG_WAVE;
  v.cap = (int64_t)UG_WAVE * S;
  char *b = (char *)ws + 256;  // first 256 B: dynamic tile counter of the shade kernel
  v.count = (int32_t *)b;
  b += ug_align256(v.n_tiles * (int64_t)sizeof(int32_t));
  v.ent = (float4 *)b;
  b += ug_align256(v.n_tiles * v.cap * (int64_t)sizeof(float4));
  v.slot = (uint8_t *)b;
  return v;
}

// blockIdx -> tile-group mapping: the dispatcher places block b on XCD (b % 8); give every XCD one
// contiguous eighth of the ray range so neighbouring image rows share that XCD's L2.
__device__ __forceinline__ int64_t ug_xcd_remap(int64_t b, int64_t nblocks) {
  const int64_t per = (nblocks + 7) / 8;
  return (b % 8) * per + b / 8;  // may be >= nblocks: caller skips
}

struct ug_march_args {
  int64_t n_rays;
  int32_t S, X, Y, Z;
  float cx, cy, cz, rx, ry, rz;        // scene centre / radius
  float lox, loy, loz, hix, hiy, hiz;  // contracted bounds
  float ex, ey, ez, irx, iry, irz;     // extent hi-lo and RN(1/extent) per axis (host computed)
  float B, A;                          // 1+bg_len, bg_len (as fp32)
  float shift, interval, thres;
};

// one density level: in-range cell set-up + one 32-byte brick + trilinear in grid_sample's order.
// `lvl` is the (wave-uniform) base of this level's bricks; the per-lane offset stays 32-bit.
__device__ __forceinline__ float ug_density_level(const char *__restrict__ lvl, float cx, float cy, float cz,
                                                  int X, int Y, int Z) {
  const ug_axis_fast ax = ug_axis_inrange(cx, X), ay = ug_axis_inrange(cy, Y), az = ug_axis_inrange(cz, Z);
  // byte offset of the cell record: the row index cx*(Y-1)+cy is exact in fp32 ((X-1)(Y-1) < 2^24), so it costs one
  // FMA + one convert; row * rowbytes is a full-rate 24-bit multiply (level < 4 GiB), then + cz*32.  The plain
  // integer form compiled to quarter-rate v_mad_u64_u32 pairs.
  const unsigned row = (unsigned)fmaf(ax.cellf, (float)(Y - 1), ay.cellf);
  const unsigned off = __umul24(row, (unsigned)(Z - 1) << 5) + ((unsigned)az.cell << 5);
  const float4 *b = (const float4 *)(lvl + off);
  const float4 v0 = b[0], v1 = b[1];
  // cell polynomial (k_pack_bricks): Horner in z, then y, then x -- 7 FMAs, no corner weights
  const float tz = az.whi, ty = ay.whi, tx = ax.whi;
  const float p00 = fmaf(v0.y, tz, v0.x), p01 = fmaf(v0.w, tz, v0.z);   // x^0: y^0, y^1
  const float p10 = fmaf(v1.y, tz, v1.x), p11 = fmaf(v1.w, tz, v1.z);   // x^1
  return fmaf(fmaf(p11, ty, p10), tx, fmaf(p01, ty, p00));
}

// DirectContractedVoxGO additions to the march (dcvgo.py:228-310; DC = true, single-level grids, F = 0): of the contracted
// samples only those are evaluated whose running inter-sample distance has just exceeded dist_thres (cumdist_thres,
// ub360_utils_kernel.cu:24-31 -- the serial recurrence is a register of the lane that owns the ray), and only where the
// mask cache (maskcache_lookup, render_utils_kernel.cu:374-392: nearest voxel of a bool grid) says "not known free space";
// wsum_mid = sum of the weights of the surviving UN-contracted samples (dcvgo.py:354-358).
struct ug_dc_args {
  const uint8_t *mask;
  int32_t mi, mj, mk;
  float sx, sy, sz, hx, hy, hz;     // xyz2ijk_scale / xyz2ijk_shift
  float dist_thres;
};

// March one 64-ray tile (lane = ray): writes alphainv_last / depth for the tile's rays, appends the
// survivors to ent/slot (this wave's private list) and returns their count (wave-uniform).
template <int F, bool L2, bool DC = false>
__device__ __forceinline__ int ug_march_tile(const ug_march_args &a, const float *__restrict__ rays_o,
                                             const float *__restrict__ rays_d, const float *__restrict__ t_table,
                                             const float *__restrict__ s_table, const float *__restrict__ bricks,
                                             float *__restrict__ alphainv_last, float *__restrict__ depth,
                                             int64_t tile, float4 *__restrict__ ent, uint8_t *__restrict__ slot,
                                             const ug_dc_args &dc = ug_dc_args{}, float *__restrict__ wsum_mid = nullptr) {
  constexpr int P = 2 * F + 1;
  const int lane = ug_lane();
  const int64_t ray = tile * UG_WAVE + lane;
  const bool valid = ray < a.n_rays;

  float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 1.f;
  if (valid) {
    const float rox = rays_o[3 * ray], roy = rays_o[3 * ray + 1], roz = rays_o[3 * ray + 2];
    const float rdx = rays_d[3 * ray], rdy = rays_d[3 * ray + 1], rdz = rays_d[3 * ray + 2];
    ox = (rox - a.cx) / a.rx; oy = (roy - a.cy) / a.ry; oz = (roz - a.cz) / a.rz;
    const float dn = ug_norm3_torch(rdx, rdy, rdz);
    dx = rdx / dn; dy = rdy / dn; dz = rdz / dn;
  }

  const size_t lvl_bytes = (size_t)(a.X - 1) * (a.Y - 1) * (a.Z - 1) * 32;  // < 4 GiB per level (G <= 512)
  const char *__restrict__ bkb = (const char *)bricks;

  float T = 1.f, dsum = 0.f;
  bool done = !valid;
  int nsurv = 0;  // wave-uniform
  [[maybe_unused]] float cum = 0.f, ppx = 0.f, ppy = 0.f, ppz = 0.f, wmid = 0.f;   // DC: cumdist recurrence, previous point

  for (int j = 0; j < a.S; ++j) {
    if (__ballot(!done) == 0ull) break;  // every ray of this wave has terminated
    bool surv = false;
    float w = 0.f;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (!done) {
      const float t = t_table[j];
      px = ox + dx * t; py = oy + dy * t; pz = oz + dz * t;
      // contraction p/|p| * (B - A/|p|) outside the unit cube / ball (FourierGrid_model.py:534-548)
      const float nrm = L2 ? ug_norm3_torch(px, py, pz) : fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz)));
      // (Markstein divisions, ugrid_math.h sin / cos / alpha, the cell polynomial: each A/B-tested against IEEE division, the device libm
      // and grid_sample's own corner sum on the S1 frame -- profiles/r02/parity_ab_s1.txt; the arms are archived, tools/experiments/ARMS.md)
      if (!(nrm <= 1.0f)) {
        const float rn = ug_rcp_refined(nrm);
        const float sc = a.B - rn * a.A;       // reciprocal(norm) * A, as torch evaluates `A / norm` (ug_contract above)
        px = ug_div_r(px, nrm, rn) * sc;
        py = ug_div_r(py, nrm, rn) * sc;
        pz = ug_div_r(pz, nrm, rn) * sc;
      }
      // ((p - lo) / (hi - lo)) * 2 - 1
      const float ux = ug_div_r(px - a.lox, a.ex, a.irx) * 2.f - 1.f;
      const float uy = ug_div_r(py - a.loy, a.ey, a.iry) * 2.f - 1.f;
      const float uz = ug_div_r(pz - a.loz, a.ez, a.irz) * 2.f - 1.f;
      bool keep = true;
      [[maybe_unused]] bool inner = true;
      if constexpr (DC) {
        inner = nrm <= 1.0f;
        keep = inner;
        if (j > 0) {      // dist = |p_j - p_{j-1}| over ALL consecutive samples (dcvgo.py:287), serial cumdist_thres recurrence
          cum += ug_norm3_torch(px - ppx, py - ppy, pz - ppz);
          const bool over = cum > dc.dist_thres;
          cum *= over ? 0.f : 1.f;
          keep = keep || over;
        }
        ppx = px; ppy = py; ppz = pz;
        if (keep) {       // mask cache: nearest voxel, C round(), NaN -> 0 like the device conversion (k_maskcache)
          float fi = roundf(px * dc.sx + dc.hx), fj = roundf(py * dc.sy + dc.hy), fk = roundf(pz * dc.sz + dc.hz);
          fi = (fi != fi) ? 0.f : fi; fj = (fj != fj) ? 0.f : fj; fk = (fk != fk) ? 0.f : fk;
          keep = false;
          if (fi >= 0.f && fi < (float)dc.mi && fj >= 0.f && fj < (float)dc.mj && fk >= 0.f && fk < (float)dc.mk)
            keep = dc.mask[((int64_t)fi * dc.mj + (int64_t)fj) * dc.mk + (int64_t)fk] != 0;
        }
      }
      if (keep) {
        float dens = ug_density_level(bkb, ux, uy, uz, a.X, a.Y, a.Z);
#pragma unroll
        for (int k = 0; k < F; ++k) {
          const float f = (float)(1 << k);
          float sx, cx_, sy, cy_, sz, cz_;
          if (k == 0) {   // |u| <= 1 < pi/2: no range reduction needed, bit-identical (ug_sincos_small)
            ug_sincos_small(ux, &sx, &cx_);
            ug_sincos_small(uy, &sy, &cy_);
            ug_sincos_small(uz, &sz, &cz_);
          } else {
            ug_sincos(f * ux, &sx, &cx_);
            ug_sincos(f * uy, &sy, &cy_);
            ug_sincos(f * uz, &sz, &cz_);
          }
          dens += ug_density_level(bkb + (size_t)(2 * k + 1) * lvl_bytes, sx, sy, sz, a.X, a.Y, a.Z);
          dens += ug_density_level(bkb + (size_t)(2 * k + 2) * lvl_bytes, cx_, cy_, cz_, a.X, a.Y, a.Z);
        }
        dens = ug_div_r(dens, (float)P, 1.0f / (float)P);   // mean over levels: Markstein division, 3 VALU instead of 10
        const float xs = dens + a.shift;
        const float alpha = ug_alpha(xs, a.interval);
        if (alpha > a.thres) {
          w = T * alpha;
          T = (float)((double)T * (1. - (double)alpha));
          if (w > a.thres) {
            surv = true;
            dsum += w * s_table[j];
            if constexpr (DC) { if (inner) wmid += w; }
          }
          if ((double)T < 1e-3) done = true;
        }
      }
    }
    const unsigned long long m = __ballot(surv);
    if (m != 0ull) {
      if (surv) {
        const int idx = nsurv + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        ent[idx] = make_float4(px, py, pz, w);
        slot[idx] = (uint8_t)lane;
      }
      nsurv += __popcll(m);
    }
  }
  if (valid) {
    alphainv_last[ray] = T;
    depth[ray] = dsum;
    if constexpr (DC) wsum_mid[ray] = wmid;
  }
  return nsurv;
}

// ----------------------------------------------------------------------------------------------
// Bounded DirectVoxGO march (dvgo.py:306-400): per-ray clipping against the scene box and VARIABLE-length marching
// (render_utils_kernel.cu: infer_t_minmax :16-41, infer_n_samples :43-57, sample_pts_on_rays :100-260) without the
// reference's count -> cumsum -> host read -> fill round trip: a lane owns a ray and simply loops to its own step count,
// the wave until its longest ray is done.  Points outside the box (mask_outbbox) and in known free space (mask cache)
// are exec-masked before the brick load; depth = sum w * step_id (dvgo.py:419-423).
// ----------------------------------------------------------------------------------------------
struct ug_dv_args {
  const uint8_t *mask;
  int32_t mi, mj, mk;
  float sx, sy, sz, hx, hy, hz;     // xyz2ijk_scale / xyz2ijk_shift
  float near, far, stepdist;
};

__device__ __forceinline__ int ug_march_tile_dvgo(const ug_march_args &a, const ug_dv_args &dv, const float *__restrict__ rays_o,
                                                  const float *__restrict__ rays_d, const float *__restrict__ bricks,
                                                  float *__restrict__ alphainv_last, float *__restrict__ depth, int64_t tile,
                                                  float4 *__restrict__ ent, uint8_t *__restrict__ slot, int cap) {
  const int lane = ug_lane();
  const int64_t ray = tile * UG_WAVE + lane;
  const bool valid = ray < a.n_rays;
  float sx = 0.f, sy = 0.f, sz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
  int n = 0;
  if (valid) {
    const float ox = rays_o[3 * ray], oy = rays_o[3 * ray + 1], oz = rays_o[3 * ray + 2];
    const float rx = rays_d[3 * ray], ry = rays_d[3 * ray + 1], rz = rays_d[3 * ray + 2];
    // ray / box slab test; a zero direction component is replaced by float(1e-6) (infer_t_minmax)
    const float vx = (rx == 0.f) ? (float)1e-6 : rx, vy = (ry == 0.f) ? (float)1e-6 : ry, vz = (rz == 0.f) ? (float)1e-6 : rz;
    const float ax = (a.hix - ox) / vx, ay = (a.hiy - oy) / vy, az = (a.hiz - oz) / vz;
    const float bx = (a.lox - ox) / vx, by = (a.loy - oy) / vy, bz = (a.loz - oz) / vz;
    const float tmin = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), dv.far), dv.near);
    const float tmax = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), dv.far), dv.near);
    const float rn = sqrtf(rx * rx + ry * ry + rz * rz);
    const double c = (double)ceilf((tmax - tmin) * rn / dv.stepdist);        // infer_n_samples
    const double nn = c > 1. ? c : 1.;
    n = nn > (double)cap ? cap : (int)nn;        // (the host sizes the work list for the box diagonal: never binding)
    sx = ox + rx * tmin; sy = oy + ry * tmin; sz = oz + rz * tmin;            // rays_start
    dx = rx / rn; dy = ry / rn; dz = rz / rn;                                  // rays_dir
  }
  const char *__restrict__ bkb = (const char *)bricks;
  float T = 1.f, dsum = 0.f;
  bool done = !valid;
  int nsurv = 0;  // wave-uniform
  for (int j = 0;; ++j) {
    done = done || j >= n;
    if (__ballot(!done) == 0ull) break;
    bool surv = false;
    float w = 0.f, px = 0.f, py = 0.f, pz = 0.f;
    if (!done) {
      const float dist = dv.stepdist * (float)j;
      px = sx + dx * dist; py = sy + dy * dist; pz = sz + dz * dist;
      bool keep = !((a.lox > px) | (a.loy > py) | (a.loz > pz) | (a.hix < px) | (a.hiy < py) | (a.hiz < pz));   // mask_outbbox
      if (keep) {       // mask cache (k_maskcache semantics)
        float fi = roundf(px * dv.sx + dv.hx), fj = roundf(py * dv.sy + dv.hy), fk = roundf(pz * dv.sz + dv.hz);
        fi = (fi != fi) ? 0.f : fi; fj = (fj != fj) ? 0.f : fj; fk = (fk != fk) ? 0.f : fk;
        keep = false;
        if (fi >= 0.f && fi < (float)dv.mi && fj >= 0.f && fj < (float)dv.mj && fk >= 0.f && fk < (float)dv.mk)
          keep = dv.mask[((int64_t)fi * dv.mj + (int64_t)fj) * dv.mk + (int64_t)fk] != 0;
      }
      if (keep) {
        const float ux = ug_div_r(px - a.lox, a.ex, a.irx) * 2.f - 1.f;
        const float uy = ug_div_r(py - a.loy, a.ey, a.iry) * 2.f - 1.f;
        const float uz = ug_div_r(pz - a.loz, a.ez, a.irz) * 2.f - 1.f;
        const float dens = ug_density_level(bkb, ux, uy, uz, a.X, a.Y, a.Z);
        const float alpha = ug_alpha(dens + a.shift, a.interval);
        if (alpha > a.thres) {
          w = T * alpha;
          T = (float)((double)T * (1. - (double)alpha));
          if (w > a.thres) {
            surv = true;
            dsum += w * (float)j;
          }
          if ((double)T < 1e-3) done = true;
        }
      }
    }
    const unsigned long long m = __ballot(surv);
    if (m != 0ull) {
      if (surv) {
        const int idx = nsurv + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        ent[idx] = make_float4(px, py, pz, w);
        slot[idx] = (uint8_t)lane;
      }
      nsurv += __popcll(m);
    }
  }
  if (valid) {
    alphainv_last[ray] = T;
    depth[ray] = dsum;
  }
  return nsurv;
}

// ----------------------------------------------------------------------------------------------
// rgbnet packing for the transposed MFMA chain
// packed (floats): A1 [KL][64][4] | A2 [64][64][4] | bias1 [2][64] | bias2 [2][64] | W3 [2][64][4] | b3 [4]
// ----------------------------------------------------------------------------------------------
__host__ __device__ static inline int ug_feat_of(int o, int r, int h) { return 32 * o + (r & 3) + 8 * (r >> 2) + 4 * h; }

// A second image of the same network follows for the bf16x3 path (each fp32 weight split into three bf16
// parts h+m+l, A operands of v_mfma_f32_32x32x16_bf16, 8 bf16 = 16 B per lane):
//   bfA1 [KB1][4 o][3 parts][64 lanes][8 bf16] | bfA2 [8][4][3][64][8] | bias1 | bias2 | W3 | b3   (fp32 tail)
//
// A third image serves the fp16x2 path (mode 2): every weight is scaled by a power of two sW (so that the
// largest |w| sits just below 2^15) and split into two fp16 parts h+l; activations are scaled by sX and split
// the same way in the kernel; three MFMA products (h.h, h.l, l.h) then carry ~2^-22 relative accuracy, i.e.
// fp32 quality at half the matrix work of bf16x3.  The tail holds the biases / W3 pre-scaled to match:
//   hxA1 [KB1][4 o][2 parts][64 lanes][8 f16] | hxA2 [8][4][2][64][8] | bias1*sW1*sX1 | bias2*sW2*sX2 |
//   W3/(sW2*sX2) | b3 | {sX1, sX2/(sW1*sX1), 0, 0}
struct ug_mlp_layout { int KL, offA1, offA2, offB1, offB2, offW3, offb3, total;
                       int KB1, bfA1, bfA2, bfB1, bfB2, bfW3, bfb3, total2;
                       int hxA1, hxA2, hxB1, hxB2, hxW3, hxb3, hxS, total3; };
// power-of-two scales of the fp16x2 image (ugrid_pack_mlp computes them on the host from the weights and the
// caller's bound on |k0|)
struct ug_mlp_scales { float sX1, sW1, sX2, sW2; };
__host__ __device__ static inline ug_mlp_layout ug_mlp_lay(int C, int n_emb) {
  const int CH = UG_CH(C);
  ug_mlp_layout L;
  L.KL = (2 * CH + n_emb + 1) / 2;
  L.offA1 = 0;
  L.offA2 = L.offA1 + L.KL * 256;
  L.offB1 = L.offA2 + 64 * 256;
  L.offB2 = L.offB1 + 128;
  L.offW3 = L.offB2 + 128;
  L.offb3 = L.offW3 + 512;
  L.total = L.offb3 + 4;
  L.KB1 = (L.KL + 7) / 8;
  L.bfA1 = L.total;                          // all offsets in floats (a bf16x8 unit = 4 floats)
  L.bfA2 = L.bfA1 + L.KB1 * 4 * 3 * 64 * 4;
  L.bfB1 = L.bfA2 + 8 * 4 * 3 * 64 * 4;
  L.bfB2 = L.bfB1 + 128;
  L.bfW3 = L.bfB2 + 128;
  L.bfb3 = L.bfW3 + 512;
  L.total2 = L.bfb3 + 4;
  L.hxA1 = L.total2;
  L.hxA2 = L.hxA1 + L.KB1 * 4 * 2 * 64 * 4;
  L.hxB1 = L.hxA2 + 8 * 4 * 2 * 64 * 4;
  L.hxB2 = L.hxB1 + 128;
  L.hxW3 = L.hxB2 + 128;
  L.hxb3 = L.hxW3 + 512;
  L.hxS = L.hxb3 + 4;
  L.total3 = L.hxS + 4;
  return L;
}

// original rgbnet input column of (step s, half h); -1 = zero padding
__host__ __device__ static inline int ug_in_col(int s, int h, int C, int n_emb, int KL) {
  const int CH = UG_CH(C);
  if (s < CH) {
    const int ch = h * CH + s;
    return ch < C ? ch : -1;
  }
  const int e = h * (KL - CH) + (s - CH);
  return e < n_emb ? C + e : -1;
}

// ----------------------------------------------------------------------------------------------
// shade
// ----------------------------------------------------------------------------------------------
#ifndef UG_K0_BATCH
#define UG_K0_BATCH 2   // k0 levels whose loads are in flight together (x 48 VGPRs at C = 12; 3 measured equal, 4 spills)
#endif

struct ug_shade_args {
  int64_t n_rays;
  int32_t X, Y, Z;
  int32_t residual;                 // 1: rgb = sigmoid(rgbnet([k0[3:], emb]) + k0[:3])  (DirectVoxGO rgbnet_direct = False, dvgo.py:385-398)
  float lox, loy, loz, hix, hiy, hiz;
  float ex, ey, ez, irx, iry, irz;  // extent hi-lo and RN(1/extent)
};

// cell set-up of one k0 level for one survivor half: record pointer + fractional coordinates
struct ug_k0_cell { const float4 *rec; float tx, ty, tz; };
template <int CH>
__device__ __forceinline__ ug_k0_cell ug_k0_cell_setup(const float *__restrict__ k0b, int h, int64_t level_base, float cx,
                                                       float cy, float cz, int X, int Y, int Z) {
  const ug_axis_fast ax = ug_axis_inrange(cx, X), ay = ug_axis_inrange(cy, Y), az = ug_axis_inrange(cz, Z);
  // row index exact in fp32 ((X-1)(Y-1) < 2^24): one FMA + one convert, then 24-bit multiplies (cells < 2^32)
  const unsigned row = (unsigned)fmaf(ax.cellf, (float)(Y - 1), ay.cellf);
  const unsigned cell = __umul24(row, (unsigned)(Z - 1)) + (unsigned)az.cell;
  ug_k0_cell c;
  c.rec = (const float4 *)(k0b + ((level_base + cell) * 2 + h) * (8 * CH));
  c.tx = ax.whi; c.ty = ay.whi; c.tz = az.whi;
  return c;
}

// k0 features of one survivor half: mean over the P levels of the cell polynomials.  The loads of NB levels are
// issued back to back before the first polynomial is evaluated (NB = 2 and 3 measured equal on MI355X: the
// shade kernel is bound by VALU + MFMA issue, not by the gather's latency).
template <int F, int CH>
__device__ __forceinline__ void ug_k0_gather(const float *__restrict__ k0b, int h, float px, float py, float pz,
                                             const ug_shade_args &a, float (&feat)[CH]) {
  constexpr int P = 2 * F + 1;
  constexpr int NB = UG_K0_BATCH;
  static_assert((8 * CH) % 4 == 0, "half-brick must be a whole number of float4");
  const float ux = ug_div_r(px - a.lox, a.ex, a.irx) * 2.f - 1.f;
  const float uy = ug_div_r(py - a.loy, a.ey, a.iry) * 2.f - 1.f;
  const float uz = ug_div_r(pz - a.loz, a.ez, a.irz) * 2.f - 1.f;
  const int64_t cells = (int64_t)(a.X - 1) * (a.Y - 1) * (a.Z - 1);
  ug_k0_cell c[P];
  c[0] = ug_k0_cell_setup<CH>(k0b, h, 0, ux, uy, uz, a.X, a.Y, a.Z);
#pragma unroll
  for (int k = 0; k < F; ++k) {
    const float f = (float)(1 << k);
    float sx, cx_, sy, cy_, sz, cz_;
    ug_sincos(f * ux, &sx, &cx_);
    ug_sincos(f * uy, &sy, &cy_);
    ug_sincos(f * uz, &sz, &cz_);
    c[2 * k + 1] = ug_k0_cell_setup<CH>(k0b, h, (int64_t)(2 * k + 1) * cells, sx, sy, sz, a.X, a.Y, a.Z);
    c[2 * k + 2] = ug_k0_cell_setup<CH>(k0b, h, (int64_t)(2 * k + 2) * cells, cx_, cy_, cz_, a.X, a.Y, a.Z);
  }
#pragma unroll
  for (int s = 0; s < CH; ++s) feat[s] = 0.f;
#pragma unroll
  for (int b0 = 0; b0 < P; b0 += NB) {
    float4 v[NB][2 * CH];
#pragma unroll
    for (int l = 0; l < NB; ++l)
      if (b0 + l < P) {
#pragma unroll
        for (int q = 0; q < 2 * CH; ++q) v[l][q] = c[b0 + l].rec[q];
      }
    __builtin_amdgcn_sched_barrier(0);   // keep every load of the batch ahead of the first use
#pragma unroll
    for (int l = 0; l < NB; ++l)
      if (b0 + l < P) {
        const float tz = c[b0 + l].tz, ty = c[b0 + l].ty, tx = c[b0 + l].tx;
#pragma unroll
        for (int pr = 0; pr < CH / 2; ++pr) {
          // half-brick layout [pair][entry][2 channels]: float4 q of the pair = entries 2q, 2q+1 x 2 channels;
          // cell polynomial by Horner in z, y, x (7 FMAs per channel)
          const float4 q0 = v[l][4 * pr], q1 = v[l][4 * pr + 1], q2 = v[l][4 * pr + 2], q3 = v[l][4 * pr + 3];
          const float a00 = fmaf(q0.z, tz, q0.x), a01 = fmaf(q1.z, tz, q1.x);
          const float a10 = fmaf(q2.z, tz, q2.x), a11 = fmaf(q3.z, tz, q3.x);
          const float b00 = fmaf(q0.w, tz, q0.y), b01 = fmaf(q1.w, tz, q1.y);
          const float b10 = fmaf(q2.w, tz, q2.y), b11 = fmaf(q3.w, tz, q3.y);
          const float fa = fmaf(fmaf(a11, ty, a10), tx, fmaf(a01, ty, a00));
          const float fb = fmaf(fmaf(b11, ty, b10), tx, fmaf(b01, ty, b00));
          feat[2 * pr] = (b0 + l == 0) ? fa : feat[2 * pr] + fa;
          feat[2 * pr + 1] = (b0 + l == 0) ? fb : feat[2 * pr + 1] + fb;
        }
      }
  }
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) feat[ch] = ug_div_r(feat[ch], (float)P, 1.0f / (float)P);
}

// ---- explicit loads ------------------------------------------------------------------------------------------
// hipcc sinks plain C++ loads down to their first use (one load in flight, s_waitcnt vmcnt(0) after each) or hoists
// all of them (spills), whatever sched_barrier says, so the k0 gather issues its loads as volatile asm (program order
// is kept among volatile asms) and waits with explicit s_waitcnt whose "+v" operands make every use of the loaded
// registers depend on the wait.  The compiler's own vmcnt bookkeeping stays safe: memory operations return in order,
// so loads it does not know about can only make its waits conservative.
typedef float ug_f4 __attribute__((ext_vector_type(4)));
template <int IMM>
__device__ __forceinline__ ug_f4 ug_gload4(unsigned voff, const float *sbase) {
  ug_f4 r;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
  return r;
}
// 64-bit address form (single-level k0 grids, F = 0: a 320^3 x 12-channel level is 12.5 GB, beyond the 32-bit offset of
// the saddr + voffset form; with P = 1 the two extra address registers per item cost nothing)
template <int IMM>
__device__ __forceinline__ ug_f4 ug_gload4v(const char *vaddr) {
  ug_f4 r;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(vaddr), "n"(IMM) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void ug_vmwait6(ug_f4 (&v)[6]) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : "n"(N));
}

// ---- quad k0 gather (C == 12) ----------------------------------------------------------------------------------
// Four ADJACENT lanes own a survivor.  Brick = [cell][q 0..5][g 0..3][4 floats] (k_pack_quad): load q of the quad reads 64
// contiguous, 64-byte aligned bytes, and lane g ends up with all 8 polynomial coefficients of channels 3g..3g+2
// (float4 2c = coefficients 0..3, 2c+1 = 4..7 of channel 3g+c).  The texture addresser works through a dwordx4 load
// four lanes (64 B) per cycle, and measured on the S1 work list (profiles/r02/gather_variants.txt) the round-1 layout
// -- lanes l / l+32 per survivor, 64 different 16-byte pieces per instruction -- ran the stand-alone gather in 5.3 ms
// against 3.4 ms for this one (same bytes, same arithmetic, bit-identical features).
template <int K>
__device__ __forceinline__ float ug_quad_bcast(float x) {   // value of lane (lane & ~3) + K: one DPP move
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), K * 0x55, 0xf, 0xf, true));
}

// 3 channels of one level from the lane's 6 float4: Horner in z, y, x (7 FMAs per channel, as ug_density_level)
__device__ __forceinline__ void ug_quad_poly(const ug_f4 (&v)[6], float tx, float ty, float tz, bool first, float (&feat)[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const ug_f4 lo = v[2 * c], hi = v[2 * c + 1];
    const float p00 = fmaf(lo.y, tz, lo.x), p01 = fmaf(lo.w, tz, lo.z);
    const float p10 = fmaf(hi.y, tz, hi.x), p11 = fmaf(hi.w, tz, hi.z);
    const float f = fmaf(fmaf(p11, ty, p10), tx, fmaf(p01, ty, p00));
    feat[c] = first ? f : feat[c] + f;
  }
}

// per-lane axis constants of the shared set-up: lane g of a quad works on axis min(g, 2) and the quad combines the
// three axes through DPP broadcasts (the set-up -- 3 sincos, 7 cell / fraction pairs -- costs a third per survivor)
struct ug_quad_axis { float lo, ex, ir, nm1, nm2; unsigned goff; };
__device__ __forceinline__ ug_quad_axis ug_quad_axis_of(const ug_shade_args &a, int g) {
  ug_quad_axis q;
  q.lo = g == 0 ? a.lox : (g == 1 ? a.loy : a.loz);
  q.ex = g == 0 ? a.ex : (g == 1 ? a.ey : a.ez);
  q.ir = g == 0 ? a.irx : (g == 1 ? a.iry : a.irz);
  const int n = g == 0 ? a.X : (g == 1 ? a.Y : a.Z);
  q.nm1 = (float)(n - 1); q.nm2 = (float)(n - 2);
  q.goff = (unsigned)g * 16u;
  return q;
}

template <int F, int NBL, int NR>
struct ug_gather_state {
  static constexpr int P = 2 * F + 1, NI = NR * P;   // pipeline items, level-major: item i = (level i / NR, round i % NR)
  unsigned off[NI];
  float tx[NI], ty[NI], tz[NI];
  ug_f4 v[NBL][6];
};

#define UG_ISSUE_ITEM(st_, i_)                                                                                     \
  if constexpr (F == 0) {   /* st.off = CELL index; 64-bit address = level 0 + cell * 384 + the lane's 16-byte column */   \
    const char *pa = (const char *)k0b + (uint64_t)st_.off[i_] * 384ull + (uint64_t)((ug_lane() & 3) * 16);        \
    st_.v[(i_) % NBL][0] = ug_gload4v<0>(pa);   st_.v[(i_) % NBL][1] = ug_gload4v<64>(pa);                          \
    st_.v[(i_) % NBL][2] = ug_gload4v<128>(pa); st_.v[(i_) % NBL][3] = ug_gload4v<192>(pa);                         \
    st_.v[(i_) % NBL][4] = ug_gload4v<256>(pa); st_.v[(i_) % NBL][5] = ug_gload4v<320>(pa);                         \
  } else {                                                                                                         \
    const float *lb = k0b + (int64_t)((i_) / NR) * lvl_floats;                                                     \
    st_.v[(i_) % NBL][0] = ug_gload4<0>(st_.off[i_], lb);   st_.v[(i_) % NBL][1] = ug_gload4<64>(st_.off[i_], lb);  \
    st_.v[(i_) % NBL][2] = ug_gload4<128>(st_.off[i_], lb); st_.v[(i_) % NBL][3] = ug_gload4<192>(st_.off[i_], lb); \
    st_.v[(i_) % NBL][4] = ug_gload4<256>(st_.off[i_], lb); st_.v[(i_) % NBL][5] = ug_gload4<320>(st_.off[i_], lb); \
  }

// first half: cell set-up of the NR survivors and the first NBL items' loads issued (nothing waited for)
template <int F, int NBL, int NR>
__device__ __forceinline__ void ug_k0_gather_begin(const float *__restrict__ k0b, const ug_shade_args &a,
                                                   const ug_quad_axis &qa, const float (&p_g)[NR],
                                                   ug_gather_state<F, NBL, NR> &st) {
  constexpr int P = 2 * F + 1, NI = NR * P;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float u = ug_div_r(p_g[r] - qa.lo, qa.ex, qa.ir) * 2.f - 1.f;
    float lc[P];
    lc[0] = u;
#pragma unroll
    for (int k = 0; k < F; ++k) {
      if constexpr (F > 0) {      // (F = 0 has no Fourier levels: lc[1] would not exist)
        if (k == 0) ug_sincos_small(u, &lc[1], &lc[2]);     // |u| <= 1: bit-identical without the range reduction
        else ug_sincos((float)(1 << k) * u, &lc[2 * k + 1], &lc[2 * k + 2]);
      }
    }
#pragma unroll
    for (int l = 0; l < P; ++l) {
      const int i = l * NR + r;
      // ug_axis_inrange with the lane's own axis length
      const float ix = fmaf(lc[l], 0.5f, 0.5f) * qa.nm1;
      const float cf = __builtin_amdgcn_fmed3f(floorf(ix), 0.0f, qa.nm2);
      const float wh = ix - cf;
      const float cxf = ug_quad_bcast<0>(cf), cyf = ug_quad_bcast<1>(cf), czf = ug_quad_bcast<2>(cf);
      st.tx[i] = ug_quad_bcast<0>(wh); st.ty[i] = ug_quad_bcast<1>(wh); st.tz[i] = ug_quad_bcast<2>(wh);
      const unsigned row = (unsigned)fmaf(cxf, (float)(a.Y - 1), cyf);     // exact in fp32: (X-1)(Y-1) < 2^24
      const unsigned cell = __umul24(row, (unsigned)(a.Z - 1)) + (unsigned)czf;
      if constexpr (F == 0) st.off[i] = cell;                               // cell index: 64-bit address formed at issue
      else st.off[i] = __umul24(cell, 384u) + qa.goff;                      // bytes inside the level (< 4 GiB)
    }
  }
  const int64_t lvl_floats = (int64_t)(a.X - 1) * (a.Y - 1) * (a.Z - 1) * 96;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NBL && i < NI; ++i) UG_ISSUE_ITEM(st, i)
  __builtin_amdgcn_sched_barrier(0);
}

// second half: the NR x P (round, level) records as ONE software pipeline with NBL items (x 6 loads) in flight -- item
// i+NBL is issued right after item i's polynomial -- so the rounds' memory latencies overlap instead of adding up.
// (Loads the compiler issues in between only make the vmcnt waits conservative: memory operations return in order.)
template <int F, int NBL, int NR>
__device__ __forceinline__ void ug_k0_gather_finish(const float *__restrict__ k0b, const ug_shade_args &a,
                                                    ug_gather_state<F, NBL, NR> &st, float (&feat)[NR][3]) {
  constexpr int P = 2 * F + 1, NI = NR * P;
  const int64_t lvl_floats = (int64_t)(a.X - 1) * (a.Y - 1) * (a.Z - 1) * 96;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int after = (NI - 1 - i) < (NBL - 1) ? (NI - 1 - i) : (NBL - 1);   // items issued after item i
    if (after == 0) ug_vmwait6<0>(st.v[i % NBL]);
    else if (after == 1) ug_vmwait6<6>(st.v[i % NBL]);
    else if (after == 2) ug_vmwait6<12>(st.v[i % NBL]);
    else if (after == 3) ug_vmwait6<18>(st.v[i % NBL]);
    else if (after == 4) ug_vmwait6<24>(st.v[i % NBL]);
    else ug_vmwait6<30>(st.v[i % NBL]);
    ug_quad_poly(st.v[i % NBL], st.tx[i], st.ty[i], st.tz[i], i < NR, feat[i % NR]);
    // keep the item's math here: without the pin the scheduler hoists every later load above it (spills)
    asm volatile("" :: "v"(feat[i % NR][0]), "v"(feat[i % NR][1]), "v"(feat[i % NR][2]));
    __builtin_amdgcn_sched_barrier(0);
    if (i + NBL < NI) { UG_ISSUE_ITEM(st, i + NBL) }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) feat[r][c] = ug_div_r(feat[r][c], (float)P, 1.0f / (float)P);
}

// ROLLING set-up variant of the same pipeline (round 4, k_shade_pc48): ug_gather_state keeps the cell set-up of ALL NR x P items
// (4 registers each: 56 at F = 3) from begin to finish; here only the level coordinates of the lane's axis (NR x P registers) stay
// alive and an item's cell offset / fractions are formed when the item is ISSUED, into a slot it shares with its load registers --
// 4 registers per item IN FLIGHT.  The 40 registers that frees are a fourth item in flight inside the 168-register budget of the
// 12-wave geometries.  Same arithmetic, same order of the polynomial sums: bit-identical features.
template <int F, int NBL, int NR>
struct ug_gather_roll {
  static constexpr int P = 2 * F + 1, NI = NR * P;
  float lc[NR][P];
  unsigned off[NBL];
  float tx[NBL], ty[NBL], tz[NBL];
  ug_f4 v[NBL][6];
};

#define UG_ROLL_SETUP_ISSUE(st_, i_)                                                                                 \
  {                                                                                                                    \
    constexpr int s_ = (i_) % NBL;                                                                                     \
    const float ix_ = fmaf(st_.lc[(i_) % NR][(i_) / NR], 0.5f, 0.5f) * qa.nm1;                                        \
    const float cf_ = __builtin_amdgcn_fmed3f(floorf(ix_), 0.0f, qa.nm2);                                              \
    const float wh_ = ix_ - cf_;                                                                                       \
    const float cxf_ = ug_quad_bcast<0>(cf_), cyf_ = ug_quad_bcast<1>(cf_), czf_ = ug_quad_bcast<2>(cf_);               \
    st_.tx[s_] = ug_quad_bcast<0>(wh_); st_.ty[s_] = ug_quad_bcast<1>(wh_); st_.tz[s_] = ug_quad_bcast<2>(wh_);         \
    const unsigned row_ = (unsigned)fmaf(cxf_, (float)(a.Y - 1), cyf_);                                                \
    const unsigned cell_ = __umul24(row_, (unsigned)(a.Z - 1)) + (unsigned)czf_;                                       \
    if constexpr (F == 0) {                                                                                            \
      st_.off[s_] = cell_;                                                                                             \
      const char *pa = (const char *)k0b + (uint64_t)cell_ * 384ull + (uint64_t)((ug_lane() & 3) * 16);                \
      st_.v[s_][0] = ug_gload4v<0>(pa);   st_.v[s_][1] = ug_gload4v<64>(pa);  st_.v[s_][2] = ug_gload4v<128>(pa);       \
      st_.v[s_][3] = ug_gload4v<192>(pa); st_.v[s_][4] = ug_gload4v<256>(pa); st_.v[s_][5] = ug_gload4v<320>(pa);       \
    } else {                                                                                                           \
      st_.off[s_] = __umul24(cell_, 384u) + qa.goff;                                                                   \
      const float *lb = k0b + (int64_t)((i_) / NR) * lvl_floats;                                                       \
      st_.v[s_][0] = ug_gload4<0>(st_.off[s_], lb);   st_.v[s_][1] = ug_gload4<64>(st_.off[s_], lb);                    \
      st_.v[s_][2] = ug_gload4<128>(st_.off[s_], lb); st_.v[s_][3] = ug_gload4<192>(st_.off[s_], lb);                   \
      st_.v[s_][4] = ug_gload4<256>(st_.off[s_], lb); st_.v[s_][5] = ug_gload4<320>(st_.off[s_], lb);                   \
    }                                                                                                                  \
  }

template <int F, int NBL, int NR, int I>
__device__ __forceinline__ void ug_roll_issue(const float *__restrict__ k0b, const ug_shade_args &a, const ug_quad_axis &qa,
                                              int64_t lvl_floats, ug_gather_roll<F, NBL, NR> &st) {
  UG_ROLL_SETUP_ISSUE(st, I)
}

template <int F, int NBL, int NR, int I>
__device__ __forceinline__ void ug_roll_step(const float *__restrict__ k0b, const ug_shade_args &a, const ug_quad_axis &qa,
                                             int64_t lvl_floats, ug_gather_roll<F, NBL, NR> &st, float (&feat)[NR][3]) {
  constexpr int NI = NR * (2 * F + 1);
  if constexpr (I < NI) {
    constexpr int after = (NI - 1 - I) < (NBL - 1) ? (NI - 1 - I) : (NBL - 1);   // items issued after item I
    ug_vmwait6<6 * after>(st.v[I % NBL]);
    ug_quad_poly(st.v[I % NBL], st.tx[I % NBL], st.ty[I % NBL], st.tz[I % NBL], I < NR, feat[I % NR]);
    asm volatile("" :: "v"(feat[I % NR][0]), "v"(feat[I % NR][1]), "v"(feat[I % NR][2]));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + NBL < NI) {
      ug_roll_issue<F, NBL, NR, I + NBL>(k0b, a, qa, lvl_floats, st);
      __builtin_amdgcn_sched_barrier(0);
    }
    ug_roll_step<F, NBL, NR, I + 1>(k0b, a, qa, lvl_floats, st, feat);
  }
}

template <int F, int NBL, int NR, int I>
__device__ __forceinline__ void ug_roll_prime(const float *__restrict__ k0b, const ug_shade_args &a, const ug_quad_axis &qa,
                                              int64_t lvl_floats, ug_gather_roll<F, NBL, NR> &st) {
  constexpr int NI = NR * (2 * F + 1);
  if constexpr (I < NBL && I < NI) {
    ug_roll_issue<F, NBL, NR, I>(k0b, a, qa, lvl_floats, st);
    ug_roll_prime<F, NBL, NR, I + 1>(k0b, a, qa, lvl_floats, st);
  }
}

template <int F, int NBL, int NR>
__device__ __forceinline__ void ug_k0_gather_quad_roll(const float *__restrict__ k0b, const ug_shade_args &a,
                                                       const ug_quad_axis &qa, const float (&p_g)[NR], float (&feat)[NR][3]) {
  constexpr int P = 2 * F + 1;
  ug_gather_roll<F, NBL, NR> st;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float u = ug_div_r(p_g[r] - qa.lo, qa.ex, qa.ir) * 2.f - 1.f;
    st.lc[r][0] = u;
#pragma unroll
    for (int k = 0; k < F; ++k) {
      if constexpr (F > 0) {
        if (k == 0) ug_sincos_small(u, &st.lc[r][1], &st.lc[r][2]);
        else ug_sincos((float)(1 << k) * u, &st.lc[r][2 * k + 1], &st.lc[r][2 * k + 2]);
      }
    }
  }
  const int64_t lvl_floats = (int64_t)(a.X - 1) * (a.Y - 1) * (a.Z - 1) * 96;
  __builtin_amdgcn_sched_barrier(0);
  ug_roll_prime<F, NBL, NR, 0>(k0b, a, qa, lvl_floats, st);
  __builtin_amdgcn_sched_barrier(0);
  ug_roll_step<F, NBL, NR, 0>(k0b, a, qa, lvl_floats, st, feat);
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) feat[r][c] = ug_div_r(feat[r][c], (float)P, 1.0f / (float)P);
}

// k0 features (mean over the P levels) of the quad's NR survivors (one per gather round) at p_g[r]: this lane's 3
// channels of each.
template <int F, int NBL, int NR>
__device__ __forceinline__ void ug_k0_gather_quad(const float *__restrict__ k0b, const ug_shade_args &a,
                                                  const ug_quad_axis &qa, const float (&p_g)[NR], float (&feat)[NR][3]) {
  ug_gather_state<F, NBL, NR> st;
  ug_k0_gather_begin<F, NBL, NR>(k0b, a, qa, p_g, st);
  ug_k0_gather_finish<F, NBL, NR>(k0b, a, st, feat);
}

__device__ __forceinline__ void ug_wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// 1/(1+exp(-x)) with a refined reciprocal (<= 1 ulp from the IEEE division, 3 VALU instead of 10); the clamp keeps
// exp finite (the Newton step would turn rcp(inf) = 0 into NaN) and changes nothing above fp32's denormal range
__device__ __forceinline__ float ug_sigmoid(float x) {
  return ug_rcp_refined(1.f + expf(-__builtin_amdgcn_fmed3f(x, -87.f, 87.f)));
}

// max(x, 0) as ONE v_max_f32: fmaxf() is preceded by a canonicalising v_max_f32 x, x, x under IEEE mode, which
// doubled the cost of the 128 ReLUs per pass (NaN in -> NaN out here, like torch.relu)
__device__ __forceinline__ float ug_relu(float x) {
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// LDS-resident packed rgbnet (A operands of the transposed MFMA chain); A1/A2 are fp32 (BF=0), bf16x8 units
// (BF=1: bf16x3) or f16x8 units (BF=2: fp16x2, with the activation scales sx1 / c12)
struct ug_mlp_lds { const float4 *A1, *A2, *W3; const float *B1, *B2, *b3; float sx1, c12; int resid; };

// four consecutive rows of the fp16x2 image's DENSE W3 ([h][64][3] floats, k_pack_mlp): rows row0 .. row0+3 of half bo / 64
// as three 16-byte reads; row0 % 4 == 0
struct ug_w3x4 { float4 a, b, c; };
__device__ __forceinline__ ug_w3x4 ug_w3_load4(const ug_mlp_lds &M, int bo, int row0) {
  const float4 *p = (const float4 *)((const float *)M.W3 + (bo + row0) * 3);
  ug_w3x4 w;
  w.a = p[0]; w.b = p[1]; w.c = p[2];
  return w;
}
// l += W3[row0 + i] * hv for i = 0..3, rows ascending (same order as the row-per-read form)
#define UG_W3_FMA4(Q_, h0_, h1_, h2_, h3_)                                                  \
  l0 = fmaf((Q_).a.x, h0_, l0); l1 = fmaf((Q_).a.y, h0_, l1); l2 = fmaf((Q_).a.z, h0_, l2);   \
  l0 = fmaf((Q_).a.w, h1_, l0); l1 = fmaf((Q_).b.x, h1_, l1); l2 = fmaf((Q_).b.y, h1_, l2);   \
  l0 = fmaf((Q_).b.z, h2_, l0); l1 = fmaf((Q_).b.w, h2_, l1); l2 = fmaf((Q_).c.x, h2_, l2);   \
  l0 = fmaf((Q_).c.y, h3_, l0); l1 = fmaf((Q_).c.z, h3_, l1); l2 = fmaf((Q_).c.w, h3_, l2)

template <int C, int PE, int BF>
__host__ __device__ static inline int ug_mlp_lds_floats() {
  const ug_mlp_layout ML = ug_mlp_lay(C, 3 + 6 * PE);
  return BF == 2 ? ML.hxS - ML.hxA1 : (BF == 1 ? ML.total2 - ML.bfA1 : ML.total);
}
// wave-private LDS scratch: [0,64) per-ray survivor bit masks + [64,192) 32 x {r,g,b,-} of the pass (ordered
// per-ray accumulation), then -- fp16x2 mode only, whose rgbnet image leaves the room -- the tile's
// view-direction embedding table [64 rays][2 halves][EH]
#define UG_ACC_SCRATCH_FLOATS 192
template <int C, int PE, int BF>
__host__ __device__ static inline int ug_wave_scratch_floats() {
  constexpr int EH = (2 * UG_CH(C) + 3 + 6 * PE + 1) / 2 - UG_CH(C);
  return BF == 2 ? UG_ACC_SCRATCH_FLOATS + 64 * 2 * EH : UG_ACC_SCRATCH_FLOATS;
}
// dynamic LDS of a shade workgroup: packed rgbnet image + one scratch per wave
template <int C, int PE, int BF, int NW>
__host__ __device__ static inline int ug_shade_lds_bytes() {
  return (int)sizeof(float) * (ug_mlp_lds_floats<C, PE, BF>() + NW * ug_wave_scratch_floats<C, PE, BF>());
}

// Residual colour (DirectVoxGO with rgbnet_direct = False, dvgo.py:385-398: the first three k0 channels are a view-independent
// "diffuse" logit added to the rgbnet's output; the rgbnet reads the remaining channels + the view embedding).  The host packs
// the first layer with ZERO columns for channels 0..2, so the MFMA chain ignores them; the passes add them to the logits.  Lane
// (h = 0, sv) holds channels 0..5 of its survivor in x[0..5]; only the h == 0 lanes publish a colour.
#define UG_RESIDUAL_ADD(M_, x_, l0_, l1_, l2_) \
  if ((M_).resid) { l0_ += (x_)[0]; l1_ += (x_)[1]; l2_ += (x_)[2]; }

template <int C, int PE, int BF>
__device__ __forceinline__ ug_mlp_lds ug_mlp_stage(float *lds, const float *__restrict__ mlp, int residual = 0) {
  const ug_mlp_layout ML = ug_mlp_lay(C, 3 + 6 * PE);
  const int base = BF == 2 ? ML.hxA1 : (BF == 1 ? ML.bfA1 : 0), n = ug_mlp_lds_floats<C, PE, BF>();
  const float4 *src = (const float4 *)(mlp + base);
  float4 *dst = (float4 *)lds;
  for (int i = threadIdx.x; i < n / 4; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  ug_mlp_lds m;
  m.A1 = (const float4 *)(lds + (BF == 2 ? ML.hxA1 : (BF == 1 ? ML.bfA1 : ML.offA1)) - base);
  m.A2 = (const float4 *)(lds + (BF == 2 ? ML.hxA2 : (BF == 1 ? ML.bfA2 : ML.offA2)) - base);
  m.B1 = lds + (BF == 2 ? ML.hxB1 : (BF == 1 ? ML.bfB1 : ML.offB1)) - base;
  m.B2 = lds + (BF == 2 ? ML.hxB2 : (BF == 1 ? ML.bfB2 : ML.offB2)) - base;
  m.W3 = (const float4 *)(lds + (BF == 2 ? ML.hxW3 : (BF == 1 ? ML.bfW3 : ML.offW3)) - base);
  m.b3 = lds + (BF == 2 ? ML.hxb3 : (BF == 1 ? ML.bfb3 : ML.offb3)) - base;
  m.sx1 = BF == 2 ? mlp[ML.hxS] : 1.f;
  m.c12 = BF == 2 ? mlp[ML.hxS + 1] : 1.f;
  m.resid = residual;
  return m;
}

// fp32 -> three bf16 parts (x = h + m + l up to 2^-24 |x|); 8 values -> one MFMA operand per part
struct ug_split3 { bf16x8 h, m, l; };
__device__ __forceinline__ ug_split3 ug_split8(const float (&x)[8]) {
  ug_split3 s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hh = (__bf16)x[i];
    const float r1 = x[i] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const float r2 = r1 - (float)mm;
    s.h[i] = hh; s.m[i] = mm; s.l[i] = (__bf16)r2;
  }
  return s;
}

// Explicit wait states around the gfx950-only bf16 MFMA.  With ROCm 7.2's hipcc the bf16x3 path produced
// run-to-run differences (one survivor's contribution perturbed in ~1 % of rays) that the fp32 MFMA path
// never shows; the pattern (timing dependent, small errors) is that of a missing VALU->MFMA / MFMA->VALU
// hazard for v_cvt_pk_bf16_f32 / v_mfma_f32_32x32x16_bf16.  These fences cost < 1 % of a pass.
__device__ __forceinline__ void ug_fence_operands() {   // around the cvt_pk that (re)builds B operands:
  __builtin_amdgcn_sched_barrier(0);                    // RAW  cvt_pk write -> MFMA read, and
  asm volatile("s_nop 15");                             // WAR  previous MFMA read -> cvt_pk write
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void ug_fence_results() {    // before VALU reads MFMA accumulators (>= 16 passes)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
  __builtin_amdgcn_sched_barrier(0);
}

// acc[o] += (Wh+Wm+Wl)(xh+xm+xl) without the three terms below 2^-24: 6 bf16 MFMAs per output tile
// (smallest terms first), issued round-robin over the 4 output tiles in a PINNED order so that an MFMA never
// reads as SrcC the accumulator written by the MFMA issued right before it.  On gfx950 / ROCm 7.2 back-to-back
// dependent v_mfma_f32_32x32x16_bf16 (which hipcc schedules freely and pads with no wait states) occasionally
// dropped the previous partial product: run-to-run differences of one survivor's contribution in ~0.1-1 % of
// rays, never seen on the fp32 MFMA path whose accumulators already rotate 4-deep.
#define UG_MFMA_BF16(acc, a, b)                                       \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);  \
  __builtin_amdgcn_sched_barrier(0)

// One k-step for the 4 output tiles.  Part-major order (Wm: xm,xh | Wl: xh | Wh: xl,xm,xh) keeps only one
// weight part (4 x bf16x8 = 16 VGPRs) live plus the prefetch of the next one: the LDS reads of the next part are
// issued before the current part's MFMAs and land while those run.  `nxt` points at the next k-step's Wm part so
// the prefetch chain continues across steps (pass nullptr-equivalent = same pointer on the last step).
struct ug_wpart { bf16x8 w[4]; };

__device__ __forceinline__ ug_wpart ug_load_part(const bf16x8 *__restrict__ Ap, int part) {
  ug_wpart p;
#pragma unroll
  for (int o = 0; o < 4; ++o) p.w[o] = Ap[(o * 3 + part) * 64];
  return p;
}

__device__ __forceinline__ void ug_mfma6x4(const bf16x8 *__restrict__ Ap, const bf16x8 *__restrict__ Ap_next,
                                           const ug_split3 &x, const float (&v_next)[8], ug_split3 &x_next,
                                           f32x16 (&acc)[4], ug_wpart &wm) {
  // wm for this step was prefetched by the caller / the previous step
  const ug_wpart wl = ug_load_part(Ap, 2);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_BF16(acc[o], wm.w[o], x.m); }
  // the matrix pipe is now busy for 4 x 32 cycles: build the NEXT k-step's bf16 operands on the VALU meanwhile
  x_next = ug_split8(v_next);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_BF16(acc[o], wm.w[o], x.h); }
  const ug_wpart wh = ug_load_part(Ap, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_BF16(acc[o], wl.w[o], x.h); }
  wm = ug_load_part(Ap_next, 1);   // next step's Wm (harmless re-read on the last step)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_BF16(acc[o], wh.w[o], x.l); }
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_BF16(acc[o], wh.w[o], x.m); }
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_BF16(acc[o], wh.w[o], x.h); }
}

// ---- fp16x2 (mode 2) -------------------------------------------------------------------------------
// fp32 -> two fp16 parts of the scaled value: x*scale = h + l up to 2^-22 |x*scale| (scale is a power of two
// chosen by the host so that |x*scale| <= 2^15, far from fp16 overflow; l only goes subnormal for
// |x*scale| < 0.25, where its absolute contribution is below 2^-26 of the layer's full scale)
struct ug_split2 { f16x8 h, l; };
// Two values per pair of v_fma_mix instructions: h = RN16(x*scale) and l = RN16(x*scale - h), each ONE fused
// operation (fp32 product, fp16 source for h, a single rounding to fp16) writing one half of the destination
// register -- 2 VALU per value instead of mul, cvt, cvt back, sub, cvt, pack.
__device__ __forceinline__ void ug_split_pair(float x0, float x1, float scale, unsigned &h, unsigned &l) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(scale));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(scale));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(scale), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(scale), "v"(h));
}
__device__ __forceinline__ ug_split2 ug_split8h(const float (&x)[8], float scale) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 hh, ll;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, l;
    ug_split_pair(x[2 * i], x[2 * i + 1], scale, h, l);
    hh[i] = h; ll[i] = l;
  }
  ug_split2 s;
  s.h = __builtin_bit_cast(f16x8, hh);
  s.l = __builtin_bit_cast(f16x8, ll);
  return s;
}

#define UG_MFMA_F16(acc, a, b)                                       \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);  \
  __builtin_amdgcn_sched_barrier(0)

struct ug_hpart { f16x8 w[4]; };
__device__ __forceinline__ ug_hpart ug_load_hpart(const f16x8 *__restrict__ Ap, int part) {
  ug_hpart p;
#pragma unroll
  for (int o = 0; o < 4; ++o) p.w[o] = Ap[(o * 2 + part) * 64];
  return p;
}

// One k-step for the 4 output tiles: acc[o] += Wl.xh + Wh.xl + Wh.xh (smallest terms first), round-robin over
// the tiles in the same pinned order as ug_mfma6x4 (no MFMA reads the accumulator written right before it).
// The low weight part of the step is prefetched by the previous step; the next step's activations are split on
// the VALU while the first four MFMAs occupy the matrix pipe.
__device__ __forceinline__ void ug_mfma3x4(const f16x8 *__restrict__ Ap, const f16x8 *__restrict__ Ap_next,
                                           const ug_split2 &x, const float (&v_next)[8], float scale,
                                           ug_split2 &x_next, f32x16 (&acc)[4], ug_hpart &wl) {
  const ug_hpart wh = ug_load_hpart(Ap, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_F16(acc[o], wl.w[o], x.h); }
  x_next = ug_split8h(v_next, scale);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_F16(acc[o], wh.w[o], x.l); }
  wl = ug_load_hpart(Ap_next, 1);   // next step's low part (harmless re-read on the last step)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) { UG_MFMA_F16(acc[o], wh.w[o], x.h); }
}

// operands of one k-step of the hand-scheduled fp16x2 pass (both weight parts, loaded during the previous step)
struct ug_kops { ug_hpart wl, wh; };

// One 32-survivor pass of the rgbnet + the ordered per-ray accumulation, shared by the classic shade tile loop
// (ug_shade_tile) and the consumer waves of the producer / consumer kernel (ugrid_shade_pc.h).  Lane (h = lane >> 5,
// sv = lane & 31) holds x[KL] = its half of survivor sv's layer-1 inputs, the survivor's weight `ww` and ray slot `sl`;
// `ok`: the survivor exists.  amask [64] / aval [32] float4: the wave's LDS scratch.  accr/accg/accb: lane = ray slot.
template <int C, int PE, int BF>
__device__ __forceinline__ void ug_rgbnet_pass(const float (&x)[(2 * UG_CH(C) + 3 + 6 * PE + 1) / 2], float ww, int sl, bool ok,
                                               const ug_mlp_lds &M, unsigned *amask, float4 *aval, float &accr, float &accg,
                                               float &accb) {
  constexpr int CH = UG_CH(C);
  constexpr int NEMB = 3 + 6 * PE;
  constexpr int KL = (2 * CH + NEMB + 1) / 2;
  const int lane = ug_lane();
  const int h = lane >> 5, sv = lane & 31;
  // ---- layers 1 and 2 on the matrix cores, transposed (H^T = W . X^T): accumulators feed the next layer
  f32x16 acc1[4], acc2[4];
  int bo = h * 64;
  asm volatile("" : "+v"(bo));  // keeps the 128 bias reads inside the pass (LICM would hoist + spill them)
  {
    const float4 *b1p = (const float4 *)(M.B1 + bo);     // 16-byte aligned: 16 ds_read_b128 instead of 32 ds_read2_b32
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = b1p[o * 4 + q];
        acc1[o][4 * q] = b.x; acc1[o][4 * q + 1] = b.y; acc1[o][4 * q + 2] = b.z; acc1[o][4 * q + 3] = b.w;
      }
  }
  if constexpr (!BF) {
    // exact fp32: v_mfma_f32_32x32x2_f32, B operand = one register (k = lane>>5 picks feature +0/+4)
#pragma unroll
    for (int s = 0; s < KL; ++s) {
      const float4 wa = M.A1[s * 64 + lane];
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.x, x[s], acc1[0], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.y, x[s], acc1[1], 0, 0, 0);
      acc1[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.z, x[s], acc1[2], 0, 0, 0);
      acc1[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.w, x[s], acc1[3], 0, 0, 0);
      if ((s & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // bound the A-operand prefetch depth
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc1[o][r] = ug_relu(acc1[o][r]);
        acc2[o][r] = ((const float4 *)(M.B2 + bo))[o * 4 + (r >> 2)][r & 3];
      }
#pragma unroll
    for (int st = 0; st < 64; ++st) {
      const float4 wa = M.A2[st * 64 + lane];
      const float xb = acc1[st >> 4][st & 15];
      acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.x, xb, acc2[0], 0, 0, 0);
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.y, xb, acc2[1], 0, 0, 0);
      acc2[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.z, xb, acc2[2], 0, 0, 0);
      acc2[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.w, xb, acc2[3], 0, 0, 0);
      if ((st & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (BF == 2) {
    // fp32-accurate through fp16x2 splitting of power-of-two-scaled operands: v_mfma_f32_32x32x16_f16, three
    // products per k-step; accumulators carry the factor sW*sX (biases / W3 are pre-scaled in the image)
    const f16x8 *A1h = (const f16x8 *)M.A1, *A2h = (const f16x8 *)M.A2;
    constexpr int KB1 = (KL + 7) / 8;
    ug_hpart wl = ug_load_hpart(A1h + lane, 1);
    ug_split2 xs, xn;
    {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (e < KL) ? x[e < KL ? e : 0] : 0.f;
      xs = ug_split8h(v, M.sx1);
    }
    ug_fence_operands();
#pragma unroll
    for (int s = 0; s < KB1; ++s) {
      float vn[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vn[e] = (8 * (s + 1) + e < KL) ? x[(8 * (s + 1) + e < KL) ? 8 * (s + 1) + e : 0] : 0.f;
      ug_mfma3x4(A1h + (s * 8) * 64 + lane, (s + 1 < KB1 ? A1h + ((s + 1) * 8) * 64 : A2h) + lane, xs, vn, M.sx1, xn, acc1, wl);
      xs = xn;
    }
    ug_fence_results();
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc1[o][r] = ug_relu(acc1[o][r]);
        acc2[o][r] = ((const float4 *)(M.B2 + bo))[o * 4 + (r >> 2)][r & 3];
      }
    {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc1[0][e];
      xs = ug_split8h(v, M.c12);
    }
    ug_fence_operands();
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      float vn[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vn[e] = acc1[(st + 1 < 8 ? st + 1 : st) >> 1][8 * ((st + 1 < 8 ? st + 1 : st) & 1) + e];
      ug_mfma3x4(A2h + (st * 8) * 64 + lane, A2h + ((st + 1 < 8 ? st + 1 : st) * 8) * 64 + lane, xs, vn, M.c12, xn, acc2, wl);
      xs = xn;
    }
    ug_fence_results();
  } else {
    // fp32-accurate through bf16x3 splitting: v_mfma_f32_32x32x16_bf16, B operand = 8 values of this lane
    // (lane half h supplies k = 8h..8h+7), i.e. 8 layer-1 inputs / 8 accumulator registers per k-step
    const bf16x8 *A1b = (const bf16x8 *)M.A1, *A2b = (const bf16x8 *)M.A2;
    constexpr int KB1 = (KL + 7) / 8;
    ug_wpart wm = ug_load_part(A1b + lane, 1);
    ug_split3 xs, xn;
    {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (e < KL) ? x[e < KL ? e : 0] : 0.f;
      xs = ug_split8(v);
    }
    ug_fence_operands();
#pragma unroll
    for (int s = 0; s < KB1; ++s) {
      float vn[8];   // inputs of the next layer-1 step (dummy zeros after the last one)
#pragma unroll
      for (int e = 0; e < 8; ++e) vn[e] = (8 * (s + 1) + e < KL) ? x[(8 * (s + 1) + e < KL) ? 8 * (s + 1) + e : 0] : 0.f;
      ug_mfma6x4(A1b + (s * 12) * 64 + lane, (s + 1 < KB1 ? A1b + ((s + 1) * 12) * 64 : A2b) + lane, xs, vn, xn, acc1, wm);
      xs = xn;
    }
    ug_fence_results();
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc1[o][r] = ug_relu(acc1[o][r]);
        acc2[o][r] = ((const float4 *)(M.B2 + bo))[o * 4 + (r >> 2)][r & 3];
      }
    {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = acc1[0][e];
      xs = ug_split8(v);
    }
    ug_fence_operands();
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      float vn[8];   // accumulator registers feeding the next k-step (re-reads the last one at the end)
      constexpr int dummy = 0; (void)dummy;
#pragma unroll
      for (int e = 0; e < 8; ++e) vn[e] = acc1[(st + 1 < 8 ? st + 1 : st) >> 1][8 * ((st + 1 < 8 ? st + 1 : st) & 1) + e];
      ug_mfma6x4(A2b + (st * 12) * 64 + lane, A2b + ((st + 1 < 8 ? st + 1 : st) * 12) * 64 + lane, xs, vn, xn, acc2, wm);
      xs = xn;
    }
    ug_fence_results();
  }
  // ---- layer 3 (3 outputs) on the VALU: each lane of the pair reduces its 64 features
  float l0 = 0.f, l1 = 0.f, l2 = 0.f;
  // W3 comes from LDS 16 rows at a time, all 16 reads issued before the first use: left to itself hipcc emits
  // read -> s_waitcnt -> 3 FMAs 64 times, one exposed LDS latency per hidden feature (phase profile: 3.3 k ticks)
  constexpr int W3B = 16;   // rows per batch
#pragma unroll
  for (int sb = 0; sb < 64; sb += W3B) {
    if constexpr (BF == 2) {      // dense W3 of the fp16x2 image: 12 reads per 16 rows
      ug_w3x4 w3[W3B / 4];
#pragma unroll
      for (int i = 0; i < W3B / 4; ++i) w3[i] = ug_w3_load4(M, bo, sb + 4 * i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < W3B / 4; ++i) {
        const int r0 = sb + 4 * i;
        const float h0 = ug_relu(acc2[r0 >> 4][r0 & 15]), h1 = ug_relu(acc2[r0 >> 4][(r0 & 15) + 1]);
        const float h2 = ug_relu(acc2[r0 >> 4][(r0 & 15) + 2]), h3 = ug_relu(acc2[r0 >> 4][(r0 & 15) + 3]);
        UG_W3_FMA4(w3[i], h0, h1, h2, h3);
      }
    } else {
      float4 w3[W3B];
#pragma unroll
      for (int i = 0; i < W3B; ++i) w3[i] = M.W3[bo + sb + i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < W3B; ++i) {
        const float hv = ug_relu(acc2[(sb + i) >> 4][(sb + i) & 15]);
        l0 = fmaf(w3[i].x, hv, l0);
        l1 = fmaf(w3[i].y, hv, l1);
        l2 = fmaf(w3[i].z, hv, l2);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  l0 = (l0 + __shfl_xor(l0, 32)) + M.b3[0];
  l1 = (l1 + __shfl_xor(l1, 32)) + M.b3[1];
  l2 = (l2 + __shfl_xor(l2, 32)) + M.b3[2];
  UG_RESIDUAL_ADD(M, x, l0, l1, l2)
  // weights.unsqueeze(-1) * rgb, then a per-ray sum in sample order (segment_coo semantics)
  const float pr = ww * ug_sigmoid(l0), pg = ww * ug_sigmoid(l1), pb = ww * ug_sigmoid(l2);
  {
    // per-ray sum in list (= sample) order through LDS: survivors publish their value and set their bit in the
    // owning ray's mask (ds_or: commutative, so deterministic); each ray lane then walks its bits upwards.
    // Each phase is closed with s_waitcnt lgkmcnt(0) + a wave barrier: with the scheduling barrier alone the fp32
    // build lost contributions on MI355X (reset / OR / read of the masks not kept in order).  The walk takes as
    // many rounds as the busiest ray has entries in the pass (1-3), against 32 readlane rounds before.
    amask[lane] = 0u;
    ug_wave_lds_sync();
    if (ok && h == 0) {
      aval[sv] = make_float4(pr, pg, pb, 0.f);
      atomicOr(&amask[sl], 1u << sv);
    }
    ug_wave_lds_sync();
    unsigned m = amask[lane];
    while (m) {
      const int k = __builtin_ctz(m);
      const float4 t = aval[k];
      accr += t.x; accg += t.y; accb += t.z;
      m &= m - 1;
    }
    __builtin_amdgcn_wave_barrier();   // the next pass rewrites aval / amask
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp16x2 rgbnet pass, hand-scheduled (UG_MLP_H2, default): same arithmetic and operation order as the BF == 2 branch of
// ug_rgbnet_pass -- results are bit-identical -- but laid out for a wave that has its SIMD's matrix pipe to itself (the
// consumer waves of the producer / consumer kernel; the classic kernel uses it too).  Phase profile of the previous
// version (profiles/r03/shade_pc_phases_*.txt): 10.4 k ticks per pass of which the 132 MFMAs need 4.2 k; the rest was VALU
// and LDS latency sitting BETWEEN MFMAs of an in-order wave:
//   * the activation split of the next k-step (8 relu + 16 v_fma_mix + the wait states between a mixlo / mixhi pair) ran
//     behind the four MFMAs of one group: ~56 cycles of VALU per 32-cycle MFMA shadow.  Now two VALU instructions follow
//     EACH of the 12 MFMAs of a k-step (relu in group 1, the high parts in group 2, the low parts in group 3; consecutive
//     v_fma_mix write different registers, so no wait state is needed);
//   * both weight parts of the next k-step are requested a whole MFMA group ahead (7-8 MFMAs of cover per ds_read_b128);
//   * layer 2's biases are read into its accumulators before layer 1 starts, layer 1's biases for the NEXT pass while
//     layer 3 runs (ug_h2_state carries those accumulators across passes) -- no bias read is waited for any more;
//   * layer 3 streams W3 from LDS in double-buffered batches of 8 rows, the first one requested before layer 2's last
//     MFMAs drain;
//   * the per-ray survivor masks are cleared by the lane that has just read them, not in a separate LDS round trip at the
//     start of the accumulation (MASK_ZEROED: the caller keeps amask for this purpose only).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef UG_MLP_H2
#define UG_MLP_H2 1
#endif
struct ug_h2_state { f32x16 acc1[4]; };     // layer-1 accumulators, pre-loaded with the layer's biases

__device__ __forceinline__ void ug_h2_preload(const ug_mlp_lds &M, int bo, ug_h2_state &st) {
  const float4 *b1p = (const float4 *)(M.B1 + bo);
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = b1p[o * 4 + q];
      st.acc1[o][4 * q] = b.x; st.acc1[o][4 * q + 1] = b.y; st.acc1[o][4 * q + 2] = b.z; st.acc1[o][4 * q + 3] = b.w;
    }
}

__device__ __forceinline__ void ug_mix_h_lo(unsigned &h, float x0, float s) { asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s)); }
__device__ __forceinline__ void ug_mix_h_hi(unsigned &h, float x1, float s) { asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s)); }
__device__ __forceinline__ void ug_mix_l_lo(unsigned &l, float x0, float s, unsigned h) {
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
}
__device__ __forceinline__ void ug_mix_l_hi(unsigned &l, float x1, float s, unsigned h) {
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
}

// one k-step, 12 MFMAs in ug_mfma3x4's order, two VALU instructions of the NEXT step's operand split behind each
template <bool RELU>
__device__ __forceinline__ void ug_mfma3x4_v3(const f16x8 *__restrict__ Ap_next, const ug_split2 &x, const float (&v_next)[8],
                                              float scale, ug_split2 &x_next, f32x16 (&acc)[4], ug_kops &k) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  float a[8];
  u32x4 hh, ll;
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  ug_hpart nwl, nwh;
  // group 1: Wl . xh   (+ relu of the next step's 8 values)
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    UG_MFMA_F16(acc[o], k.wl.w[o], x.h);
    a[2 * o] = RELU ? ug_relu(v_next[2 * o]) : v_next[2 * o];
    a[2 * o + 1] = RELU ? ug_relu(v_next[2 * o + 1]) : v_next[2 * o + 1];
    if (RELU) asm volatile("" :: "v"(a[2 * o]), "v"(a[2 * o + 1]));
    __builtin_amdgcn_sched_barrier(0);
  }
  // group 2: Wh . xl   (+ high parts; the next step's low weights are requested behind the first MFMA)
  UG_MFMA_F16(acc[0], k.wh.w[0], x.l); ug_mix_h_lo(h0, a[0], scale); ug_mix_h_lo(h1, a[2], scale);
  nwl = ug_load_hpart(Ap_next, 1); __builtin_amdgcn_sched_barrier(0);
  UG_MFMA_F16(acc[1], k.wh.w[1], x.l); ug_mix_h_hi(h0, a[1], scale); ug_mix_h_hi(h1, a[3], scale); __builtin_amdgcn_sched_barrier(0);
  UG_MFMA_F16(acc[2], k.wh.w[2], x.l); ug_mix_h_lo(h2, a[4], scale); ug_mix_h_lo(h3, a[6], scale); __builtin_amdgcn_sched_barrier(0);
  UG_MFMA_F16(acc[3], k.wh.w[3], x.l); ug_mix_h_hi(h2, a[5], scale); ug_mix_h_hi(h3, a[7], scale); __builtin_amdgcn_sched_barrier(0);
  // group 3: Wh . xh   (+ low parts; the next step's high weights behind the first MFMA)
  UG_MFMA_F16(acc[0], k.wh.w[0], x.h); ug_mix_l_lo(l0, a[0], scale, h0); ug_mix_l_lo(l1, a[2], scale, h1);
  nwh = ug_load_hpart(Ap_next, 0); __builtin_amdgcn_sched_barrier(0);
  UG_MFMA_F16(acc[1], k.wh.w[1], x.h); ug_mix_l_hi(l0, a[1], scale, h0); ug_mix_l_hi(l1, a[3], scale, h1); __builtin_amdgcn_sched_barrier(0);
  UG_MFMA_F16(acc[2], k.wh.w[2], x.h); ug_mix_l_lo(l2, a[4], scale, h2); ug_mix_l_lo(l3, a[6], scale, h3); __builtin_amdgcn_sched_barrier(0);
  UG_MFMA_F16(acc[3], k.wh.w[3], x.h); ug_mix_l_hi(l2, a[5], scale, h2); ug_mix_l_hi(l3, a[7], scale, h3); __builtin_amdgcn_sched_barrier(0);
  hh[0] = h0; hh[1] = h1; hh[2] = h2; hh[3] = h3;
  ll[0] = l0; ll[1] = l1; ll[2] = l2; ll[3] = l3;
  x_next.h = __builtin_bit_cast(f16x8, hh);
  x_next.l = __builtin_bit_cast(f16x8, ll);
  k.wl = nwl; k.wh = nwh;
}

// CARRY: the caller's state holds the layer-1 accumulators (pre-loaded with the biases) across passes -- a consumer wave has
// the registers for that; the classic kernel, whose gather needs them, loads the biases at the start of the pass instead.
template <int C, int PE, bool MASK_ZEROED, bool CARRY>
__device__ __forceinline__ void ug_rgbnet_pass_h2(const float (&x)[(2 * UG_CH(C) + 3 + 6 * PE + 1) / 2], float ww, int sl, bool ok,
                                                  const ug_mlp_lds &M, unsigned *amask, float4 *aval, float &accr, float &accg,
                                                  float &accb, ug_h2_state &st) {
  constexpr int CH = UG_CH(C);
  constexpr int NEMB = 3 + 6 * PE;
  constexpr int KL = (2 * CH + NEMB + 1) / 2;
  constexpr int KB1 = (KL + 7) / 8;
  const int lane = ug_lane();
  const int h = lane >> 5, sv = lane & 31;
  int bo = h * 64;
  asm volatile("" : "+v"(bo));  // keeps the bias / W3 reads inside the pass (LICM would hoist + spill them)
  f32x16 acc2[4];
  const f16x8 *A1h = (const f16x8 *)M.A1, *A2h = (const f16x8 *)M.A2;
  if constexpr (!CARRY) ug_h2_preload(M, bo, st);
  // layer 2's biases: straight into its accumulators, landing while layer 1 runs
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = ((const float4 *)(M.B2 + bo))[o * 4 + q];
      acc2[o][4 * q] = b.x; acc2[o][4 * q + 1] = b.y; acc2[o][4 * q + 2] = b.z; acc2[o][4 * q + 3] = b.w;
    }
  ug_kops kop;
  kop.wl = ug_load_hpart(A1h + lane, 1);
  kop.wh = ug_load_hpart(A1h + lane, 0);
  ug_split2 xs, xn;
  {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (e < KL) ? x[e < KL ? e : 0] : 0.f;
    xs = ug_split8h(v, M.sx1);
  }
  ug_fence_operands();
#pragma unroll
  for (int s = 0; s < KB1; ++s) {
    float vn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) vn[e] = (8 * (s + 1) + e < KL) ? x[(8 * (s + 1) + e < KL) ? 8 * (s + 1) + e : 0] : 0.f;
    ug_mfma3x4_v3<false>((s + 1 < KB1 ? A1h + ((s + 1) * 8) * 64 : A2h) + lane, xs, vn, M.sx1, xn, st.acc1, kop);
    xs = xn;
  }
  ug_fence_results();
  {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ug_relu(st.acc1[0][e]);
    xs = ug_split8h(v, M.c12);
  }
  ug_fence_operands();
  constexpr int W3B = 8;        // W3 rows per batch, two batches in registers (dense image: 6 ds_read_b128 per batch)
  ug_w3x4 w3[2][W3B / 4];
#pragma unroll
  for (int st8 = 0; st8 < 8; ++st8) {
    float vn[8];
    const int nx = st8 + 1 < 8 ? st8 + 1 : st8;
#pragma unroll
    for (int e = 0; e < 8; ++e) vn[e] = st.acc1[nx >> 1][8 * (nx & 1) + e];
    ug_mfma3x4_v3<true>(A2h + (nx * 8) * 64 + lane, xs, vn, M.c12, xn, acc2, kop);
    xs = xn;
  }
  // first batch of layer 3's weights: requested before layer 2's last MFMAs have drained
#pragma unroll
  for (int i = 0; i < W3B / 4; ++i) w3[0][i] = ug_w3_load4(M, bo, 4 * i);
  ug_fence_results();
  // the NEXT pass's layer-1 biases into the (now dead) layer-1 accumulators: they land while layer 3 runs
  if constexpr (CARRY) ug_h2_preload(M, bo, st);
  // ---- layer 3 (3 outputs) on the VALU: each lane of the pair reduces its 64 hidden features
  float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
  for (int sb = 0; sb < 64; sb += W3B) {
    const int cur = (sb / W3B) & 1;
    if (sb + W3B < 64) {
#pragma unroll
      for (int i = 0; i < W3B / 4; ++i) w3[cur ^ 1][i] = ug_w3_load4(M, bo, sb + W3B + 4 * i);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < W3B / 4; ++i) {
      const int r0 = sb + 4 * i;
      const float h0 = ug_relu(acc2[r0 >> 4][r0 & 15]), h1 = ug_relu(acc2[r0 >> 4][(r0 & 15) + 1]);
      const float h2 = ug_relu(acc2[r0 >> 4][(r0 & 15) + 2]), h3 = ug_relu(acc2[r0 >> 4][(r0 & 15) + 3]);
      UG_W3_FMA4(w3[cur][i], h0, h1, h2, h3);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  l0 = (l0 + __shfl_xor(l0, 32)) + M.b3[0];
  l1 = (l1 + __shfl_xor(l1, 32)) + M.b3[1];
  l2 = (l2 + __shfl_xor(l2, 32)) + M.b3[2];
  UG_RESIDUAL_ADD(M, x, l0, l1, l2)
  const float pr = ww * ug_sigmoid(l0), pg = ww * ug_sigmoid(l1), pb = ww * ug_sigmoid(l2);
  {
    // ordered per-ray sum through LDS, as in ug_rgbnet_pass
    if constexpr (!MASK_ZEROED) {
      amask[lane] = 0u;
      ug_wave_lds_sync();
    }
    if (ok && h == 0) {
      aval[sv] = make_float4(pr, pg, pb, 0.f);
      atomicOr(&amask[sl], 1u << sv);
    }
    ug_wave_lds_sync();
    unsigned m = amask[lane];
    if constexpr (MASK_ZEROED) amask[lane] = 0u;     // ready for the next pass (LDS operations of a wave execute in order)
    while (m) {
      const int kk = __builtin_ctz(m);
      const float4 t = aval[kk];
      accr += t.x; accg += t.y; accb += t.z;
      m &= m - 1;
    }
    __builtin_amdgcn_wave_barrier();   // the next pass rewrites aval / amask
  }
}

// Shade one tile's survivor list (32 survivors per pass, lanes l / l+32 pair up) and write the tile's
// rgb_marched.  C = 2*CH or 2*CH-1 k0 channels, PE view-direction frequencies; rgbnet 128 wide, 3 layers.
// C == 12: k0 bricks in the quad layout, gathered 16 survivors at a time by lane quads and transposed into the MFMA
// operand layout through 768 B of the wave's LDS scratch; other C: pair half-bricks gathered directly in that layout.
template <int F, int C, int PE, int BF>
__device__ __forceinline__ void ug_shade_tile(const ug_shade_args &a, const float *__restrict__ viewdirs,
                                              const float *__restrict__ k0b, const ug_mlp_lds &M, int64_t tile,
                                              int count, const float4 *__restrict__ ent,
                                              const uint8_t *__restrict__ slot,
                                              float *__restrict__ scr, float *__restrict__ rgb_marched) {
  constexpr int CH = UG_CH(C);
  constexpr int NEMB = 3 + 6 * PE;
  constexpr int KL = (2 * CH + NEMB + 1) / 2;
  const int lane = ug_lane();
  const int h = lane >> 5, sv = lane & 31;
  float accr = 0.f, accg = 0.f, accb = 0.f;  // lane = ray slot of this tile
  constexpr int EH = KL - CH;                // embedding values per lane half
  constexpr bool EMB_LDS = (BF == 2);
  unsigned *amask = (unsigned *)scr;         // [64] bit k set: entry k of the pass belongs to this ray slot
  float4 *aval = (float4 *)(scr + 64);       // [32] weighted rgb of entry k
  float *embt = scr + UG_ACC_SCRATCH_FLOATS; // [64][2][EH]
  if constexpr (EMB_LDS) {
    // the embedding depends on the ray only: build it once per tile (lane = ray slot) instead of once per
    // survivor (a tile averages ~25 passes in S1), 12 sincos + 3 global loads per pass saved
    int64_t ray = tile * UG_WAVE + lane;
    if (ray >= a.n_rays) ray = a.n_rays - 1;
    const float vx = viewdirs[3 * ray], vy = viewdirs[3 * ray + 1], vz = viewdirs[3 * ray + 2];
    float emb[2 * EH];
    emb[0] = vx; emb[1] = vy; emb[2] = vz;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const float v = ax == 0 ? vx : (ax == 1 ? vy : vz);
#pragma unroll
      for (int k = 0; k < PE; ++k) {
        float s_, c_;
        ug_sincos(v * (float)(1 << k), &s_, &c_);
        emb[3 + ax * PE + k] = s_;
        emb[3 + 3 * PE + ax * PE + k] = c_;
      }
    }
#pragma unroll
    for (int e = NEMB; e < 2 * EH; ++e) emb[e] = 0.f;
    __builtin_amdgcn_wave_barrier();   // the previous tile's last pass has read its table rows
#pragma unroll
    for (int e = 0; e < 2 * EH; ++e) embt[lane * (2 * EH) + e] = emb[e];
    ug_wave_lds_sync();
  }

  constexpr bool QUAD = (C == 12);
  const int qs = lane >> 2, qg = lane & 3;
  const ug_quad_axis qa = ug_quad_axis_of(a, qg);
  // the work-list entries of the NEXT pass are fetched while this pass's rgbnet runs (software prefetch): the lane's own
  // survivor (weight, ray slot) and, in the quad layout, the positions of the two survivors its quad gathers
  float w_n = 0.f, pg0_n = 0.f, pg1_n = 0.f;
  float4 en_n = make_float4(0.f, 0.f, 0.f, 0.f);
  int sl_n = 0;
  if (sv < count) { sl_n = slot[sv]; if constexpr (QUAD) w_n = ent[sv].w; else en_n = ent[sv]; }
  if constexpr (QUAD) {
    const float *ef = (const float *)ent;
    if (qs < count) pg0_n = ef[4 * qs + (qg < 2 ? qg : 2)];
    if (16 + qs < count) pg1_n = ef[4 * (16 + qs) + (qg < 2 ? qg : 2)];
  }
  ug_gather_state<F, 4, 2> gst;
  for (int base = 0; base < count; base += 32) {
    const int e = base + sv;
    const bool ok = e < count;
    const float4 en = QUAD ? make_float4(0.f, 0.f, 0.f, w_n) : en_n;
    const int sl = sl_n;
    const float pg0 = pg0_n, pg1 = pg1_n;
    {
      const int e2 = e + 32;
      en_n = make_float4(0.f, 0.f, 0.f, 0.f);
      sl_n = 0; w_n = 0.f; pg0_n = 0.f; pg1_n = 0.f;
      if (e2 < count) { sl_n = slot[e2]; if constexpr (QUAD) w_n = ent[e2].w; else en_n = ent[e2]; }
      if constexpr (QUAD) {
        const float *ef = (const float *)ent;
        const int q0 = base + 32 + qs, q1 = base + 48 + qs;
        if (q0 < count) pg0_n = ef[4 * q0 + (qg < 2 ? qg : 2)];
        if (q1 < count) pg1_n = ef[4 * q1 + (qg < 2 ? qg : 2)];
      }
    }
    // ---- layer-1 inputs of this lane: half of k0 + half of the view-direction embedding
    float x[KL];
    {
      float feat[CH];
      if constexpr (QUAD) {
        // two rounds of 16 survivors (one per quad); lane (qs, qg) computes channels 3qg..3qg+2, the round's 16 x 12
        // features go through the scratch ([survivor][12], conflict-free both ways) and lane (h, sv) of that round
        // picks its 6 channels.  LDS operations of a wave execute in order: write -> read -> next round's write.
        float *xp = scr;
        float f3[2][3];
        const float pgs[2] = {pg0, pg1};
        ug_k0_gather_begin<F, 4, 2>(k0b, a, qa, pgs, gst);
        ug_k0_gather_finish<F, 4, 2>(k0b, a, gst, f3);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          xp[qs * 12 + 3 * qg + 0] = f3[it][0]; xp[qs * 12 + 3 * qg + 1] = f3[it][1]; xp[qs * 12 + 3 * qg + 2] = f3[it][2];
          ug_wave_lds_sync();
          const float *rp = xp + (sv & 15) * 12 + h * 6;
          const bool mine = ((sv >> 4) == it);
#pragma unroll
          for (int k = 0; k < 6; ++k) { const float t = rp[k]; feat[k] = (it == 0 || mine) ? t : feat[k]; }
          ug_wave_lds_sync();
        }
      } else {
        ug_k0_gather<F, CH>(k0b, h, en.x, en.y, en.z, a, feat);
      }
#pragma unroll
      for (int s = 0; s < CH; ++s) x[s] = (h * CH + s < C) ? feat[s] : 0.f;
      if constexpr (EMB_LDS) {
        const float *er = embt + sl * (2 * EH) + h * EH;
#pragma unroll
        for (int s = CH; s < KL; ++s) x[s] = er[s - CH];
      } else {
        int64_t ray = tile * UG_WAVE + sl;
        if (ray >= a.n_rays) ray = a.n_rays - 1;
        const float vx = viewdirs[3 * ray], vy = viewdirs[3 * ray + 1], vz = viewdirs[3 * ray + 2];
        float emb[NEMB];
        emb[0] = vx; emb[1] = vy; emb[2] = vz;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const float v = ax == 0 ? vx : (ax == 1 ? vy : vz);
#pragma unroll
          for (int k = 0; k < PE; ++k) {
            float s_, c_;
            ug_sincos(v * (float)(1 << k), &s_, &c_);
            emb[3 + ax * PE + k] = s_;
            emb[3 + 3 * PE + ax * PE + k] = c_;
          }
        }
#pragma unroll
        for (int s = CH; s < KL; ++s) {
          const int e0 = s - CH, e1 = (KL - CH) + (s - CH);
          const float lo = emb[e0];
          const float hi = (e1 < NEMB) ? emb[e1 < NEMB ? e1 : 0] : 0.f;
          x[s] = h ? hi : lo;
        }
      }
    }
    if constexpr (BF == 2 && UG_MLP_H2) {
      ug_h2_state h2st;
      ug_rgbnet_pass_h2<C, PE, false, false>(x, en.w, sl, ok, M, amask, aval, accr, accg, accb, h2st);
    }
    else ug_rgbnet_pass<C, PE, BF>(x, en.w, sl, ok, M, amask, aval, accr, accg, accb);
  }
  const int64_t ray = tile * UG_WAVE + lane;
  if (ray < a.n_rays) {
    rgb_marched[3 * ray] = accr;
    rgb_marched[3 * ray + 1] = accg;
    rgb_marched[3 * ray + 2] = accb;
  }
}



// dynamic tile scheduling with XCD affinity: the tile range is cut into 8 contiguous eighths, one atomic
// counter each; a workgroup (XCD = blockIdx % 8) drains its own eighth first, then steals from the others.
// Placement only affects speed, never results.  Returns -1 when every eighth is exhausted.
__device__ __forceinline__ int64_t ug_next_tile(int32_t *__restrict__ tile_counter, int64_t n_tiles, int home,
                                                int &victim) {
  const int64_t per = (n_tiles + 7) / 8;
  const int lane = ug_lane();
  while (victim < 8) {
    const int q = (home + victim) & 7;
    int t = 0;
    if (lane == 0) t = atomicAdd(tile_counter + q, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    const int64_t cand = (int64_t)q * per + t;
    if (t < per && cand < n_tiles) return cand;
    ++victim;
  }
  return -1;
}


#define ST(s) ((hipStream_t)(s))

static inline int ug_fill_march_args(const ugrid_render_params *p, ug_march_args &a) {
  if (p->n_samples <= 0 || p->grid_x < 2 || p->grid_y < 2 || p->grid_z < 2) return (int)hipErrorInvalidValue;
  if ((int64_t)(p->grid_x - 1) * (p->grid_y - 1) * (p->grid_z - 1) * 32 >= ((int64_t)1 << 32)) return (int)hipErrorInvalidValue;
  a.n_rays = p->n_rays; a.S = p->n_samples; a.X = p->grid_x; a.Y = p->grid_y; a.Z = p->grid_z;
  a.cx = p->scene_center[0]; a.cy = p->scene_center[1]; a.cz = p->scene_center[2];
  a.rx = p->scene_radius[0]; a.ry = p->scene_radius[1]; a.rz = p->scene_radius[2];
  a.lox = p->xyz_min[0]; a.loy = p->xyz_min[1]; a.loz = p->xyz_min[2];
  a.hix = p->xyz_max[0]; a.hiy = p->xyz_max[1]; a.hiz = p->xyz_max[2];
  a.ex = a.hix - a.lox; a.ey = a.hiy - a.loy; a.ez = a.hiz - a.loz;   // fp32, like (xyz_max - xyz_min)
  a.irx = 1.0f / a.ex; a.iry = 1.0f / a.ey; a.irz = 1.0f / a.ez;       // IEEE RN(1/extent)
  // python: B = 1 + bg_len, A = B*1 - 1 (doubles) then cast to fp32 when they meet the tensor
  const double Bd = 1.0 + (double)p->bg_len;
  a.B = (float)Bd; a.A = (float)(Bd * 1.0 - 1.0);
  a.shift = p->act_shift; a.interval = p->interval; a.thres = p->thres;
  return 0;
}

// Zero a few 32-bit words on the stream with a KERNEL instead of hipMemsetAsync: the shade kernels' tile counters are hit by
// device-scope atomics and live in L2; a memset NODE of a captured hipGraph does not reach them coherently on replay (ROCm 7.2,
// MI355X: from the second replay on every workgroup saw its counter exhausted and 3 of 4 tiles kept the previous frame's colours
// -- tests/test_gpu_fused.py::test_frame_render_is_capturable_in_a_hip_graph), a kernel's stores do.  ~2 us per launch.
static __global__ void ug_k_zero_u32(uint32_t *__restrict__ p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0u;
}
#define UG_ZERO_WORDS(ptr, n_words, st)                                                                      \
  do {                                                                                                       \
    hipLaunchKernelGGL(ug_k_zero_u32, dim3(1), dim3(64), 0, (st), (uint32_t *)(ptr), (int)(n_words));        \
    UG_LAUNCH_CHECK();                                                                                       \
  } while (0)

static inline void ug_fill_shade_args(const ugrid_render_params *p, ug_shade_args &a) {
  a.n_rays = p->n_rays; a.X = p->grid_x; a.Y = p->grid_y; a.Z = p->grid_z;
  a.lox = p->xyz_min[0]; a.loy = p->xyz_min[1]; a.loz = p->xyz_min[2];
  a.hix = p->xyz_max[0]; a.hiy = p->xyz_max[1]; a.hiz = p->xyz_max[2];
  a.ex = a.hix - a.lox; a.ey = a.hiy - a.loy; a.ez = a.hiz - a.loz;
  a.irx = 1.0f / a.ex; a.iry = 1.0f / a.ey; a.irz = 1.0f / a.ez;
  a.residual = (p->mlp_mode & UGRID_MLP_RESIDUAL) ? 1 : 0;
}

