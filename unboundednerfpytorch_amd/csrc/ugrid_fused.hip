// libugrid_hip.so -- fused FourierGrid render path for gfx950 (MI355X).
//
// Replaces, for inference, the torch-op chain of the reference's FourierGridModel.forward
// (FourierGrid/FourierGrid_model.py:509-672) and FourierGrid.forward (FourierGrid_grid.py:60-78):
//
//   k_march : 1 lane = 1 ray, 1 wave = 64 consecutive rays.  Per sample: contraction, Fourier
//             level coordinates, ONE 32-byte brick load per level (the 2x2x2 neighbourhood of the
//             trilinear cell, see DESIGN.md "brick layout"), mean over levels, raw2alpha, the two
//             thresholds and the front-to-back transmittance recurrence -- which is a plain serial
//             multiply in the lane's registers because a lane owns a ray.  A wave leaves the sample
//             loop as soon as all of its 64 rays have terminated (T < 1e-3) -- wave-level early
//             termination by ballot.  Surviving samples are compacted per wave (ballot + mbcnt
//             prefix) into that wave's private slice of the work list: no atomics, deterministic.
//   k_shade : 1 wave walks one tile's survivor list 32 at a time.  Lanes l and l+32 form a pair
//             that owns survivor (l&31): each gathers half of the k0 channels from the 2x2x2 k0
//             bricks and half of the view-direction embedding, which makes their registers exactly
//             the B operand of v_mfma_f32_32x32x2_f32 (B[k=l>>5][j=l&31]).  The rgbnet runs
//             "transposed" (H^T = W . X^T) so every layer's accumulator registers are directly the
//             next layer's B operands -- activations never leave the register file; packed weights
//             (A operands) are read from LDS.  fp32-input MFMA is bit-wise an fmaf chain, so the
//             MLP stays inside the 1e-4 parity budget (no bf16 anywhere).
//
// Compiled with -ffp-contract=off: every a*b+c below that must match torch's separate
// multiply/add is written as such; fmaf is explicit where torch's CPU kernels use FMA.
#include "ugrid_common.h"
#include "ugrid_math.h"
#include <string.h>

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

// FourierGrid_model.py:534-548: p/|p| * ((1+bg) - bg/|p|) outside the unit cube (inf) / ball (l2)
template <bool L2>
__device__ __forceinline__ ug_vec3 ug_contract(ug_vec3 p, float B, float A) {
  const float nrm = L2 ? ug_norm3_torch(p.x, p.y, p.z) : fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z)));
  if (!(nrm <= 1.0f)) {
    const float sc = B - A / nrm;
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
// brick packing: canonical [P,C,X,Y,Z] -> [P*(X-1)(Y-1)(Z-1)] records of [H halves][8 corners][CH]
// ----------------------------------------------------------------------------------------------
__global__ void k_pack_bricks(const float *__restrict__ grid, int P, int C, int X, int Y, int Z, int H,
                              int CH, float *__restrict__ out, int64_t total) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total;
       o += (int64_t)gridDim.x * blockDim.x) {
    int64_t q = o;
    const int ch = (int)(q % CH); q /= CH;
    const int c = (int)(q % 8); q /= 8;
    const int h = (int)(q % H); q /= H;
    const int k = (int)(q % (Z - 1)); q /= (Z - 1);
    const int j = (int)(q % (Y - 1)); q /= (Y - 1);
    const int i = (int)(q % (X - 1)); q /= (X - 1);
    const int l = (int)q;
    const int chan = h * CH + ch;
    float v = 0.f;
    if (chan < C) {
      const int ii = i + (c >> 2), jj = j + ((c >> 1) & 1), kk = k + (c & 1);
      v = grid[((((int64_t)l * C + chan) * X + ii) * Y + jj) * Z + kk];
    }
    out[o] = v;
  }
}

// ----------------------------------------------------------------------------------------------
// stand-alone grid query on the canonical layout (FourierGrid.forward / DenseGrid.forward).
// 1 lane per point; corner taps are z-pairs in the [.., Z] fastest dimension.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float ug_tap(const float *__restrict__ g, int X, int Y, int Z, float cx, float cy,
                                        float cz) {
  // generic zero-padded trilinear tap at normalised (cx->X axis, cy->Y, cz->Z)
  const float ix = ((cx + 1.f) / 2.f) * (float)(X - 1);
  const float iy = ((cy + 1.f) / 2.f) * (float)(Y - 1);
  const float iz = ((cz + 1.f) / 2.f) * (float)(Z - 1);
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  const float wx0 = (fx + 1.f) - ix, wx1 = ix - fx;
  const float wy0 = (fy + 1.f) - iy, wy1 = iy - fy;
  const float wz0 = (fz + 1.f) - iz, wz1 = iz - fz;
  const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)X), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Y),
            z0 = (int)fminf(fmaxf(fz, -2.f), (float)Z);
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int xi = x0 + (c >> 2), yi = y0 + ((c >> 1) & 1), zi = z0 + (c & 1);
    if (xi >= 0 && xi < X && yi >= 0 && yi < Y && zi >= 0 && zi < Z) {
      const float w = ((c & 1) ? wz1 : wz0) * (((c >> 1) & 1) ? wy1 : wy0) * ((c >> 2) ? wx1 : wx0);
      acc += g[((int64_t)xi * Y + yi) * Z + zi] * w;
    }
  }
  return acc;
}

__global__ void k_grid_query(const float *__restrict__ grid, int P, int C, int X, int Y, int Z,
                             const float *__restrict__ xyz, const float *__restrict__ xyz_min,
                             const float *__restrict__ xyz_max, int F, int64_t n, float *__restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float ux = ug_unorm(xyz[3 * p], xyz_min[0], xyz_max[0]);
  const float uy = ug_unorm(xyz[3 * p + 1], xyz_min[1], xyz_max[1]);
  const float uz = ug_unorm(xyz[3 * p + 2], xyz_min[2], xyz_max[2]);
  const int64_t vol = (int64_t)X * Y * Z;
  for (int ch = 0; ch < C; ++ch) {
    float acc = 0.f;
    for (int l = 0; l < P; ++l) {
      float cx = ux, cy = uy, cz = uz;
      if (l > 0) {
        const float f = (float)(1 << ((l - 1) >> 1));
        if ((l - 1) & 1) { cx = cosf(f * ux); cy = cosf(f * uy); cz = cosf(f * uz); }
        else { cx = sinf(f * ux); cy = sinf(f * uy); cz = sinf(f * uz); }
      }
      const float v = ug_tap(grid + ((int64_t)l * C + ch) * vol, X, Y, Z, cx, cy, cz);
      acc = (l == 0) ? v : acc + v;
    }
    out[p * C + ch] = (F > 0) ? acc / (float)P : acc;
  }
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

extern "C" int64_t ugrid_render_ws_bytes(int64_t n_rays, int32_t S) {
  const int64_t n_tiles = (n_rays + UG_WAVE - 1) / UG_WAVE, cap = (int64_t)UG_WAVE * S;
  return 256 + ug_align256(n_tiles * 4) + ug_align256(n_tiles * cap * 16) + ug_align256(n_tiles * cap);
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
  const unsigned rec = ((unsigned)ax.cell * (unsigned)(Y - 1) + (unsigned)ay.cell) * (unsigned)(Z - 1) + (unsigned)az.cell;
  const float4 *b = (const float4 *)(lvl + (size_t)(rec * 32u));
  const float4 v0 = b[0], v1 = b[1];
  const float w00 = az.wlo * ay.wlo, w01 = az.whi * ay.wlo, w10 = az.wlo * ay.whi, w11 = az.whi * ay.whi;
  float acc = v0.x * (w00 * ax.wlo);
  acc += v0.y * (w01 * ax.wlo);
  acc += v0.z * (w10 * ax.wlo);
  acc += v0.w * (w11 * ax.wlo);
  acc += v1.x * (w00 * ax.whi);
  acc += v1.y * (w01 * ax.whi);
  acc += v1.z * (w10 * ax.whi);
  acc += v1.w * (w11 * ax.whi);
  return acc;
}

template <int F, bool L2>
__global__ void __launch_bounds__(256)
k_march(ug_march_args a, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
        const float *__restrict__ t_table, const float *__restrict__ s_table,
        const float *__restrict__ bricks, float *__restrict__ alphainv_last, float *__restrict__ depth,
        ug_ws_view ws, int64_t nblocks) {
  constexpr int P = 2 * F + 1;
  const int64_t blk = ug_xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int lane = ug_lane();
  const int64_t tile = blk * 4 + (threadIdx.x >> 6);
  if (tile >= ws.n_tiles) return;
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
  float4 *__restrict__ ent = ws.ent + tile * ws.cap;
  uint8_t *__restrict__ slot = ws.slot + tile * ws.cap;

  float T = 1.f, dsum = 0.f;
  bool done = !valid;
  int nsurv = 0;  // wave-uniform

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
      if (!(nrm <= 1.0f)) {
        const float rn = ug_rcp_refined(nrm);
        const float sc = a.B - ug_div_r(a.A, nrm, rn);
        px = ug_div_r(px, nrm, rn) * sc;
        py = ug_div_r(py, nrm, rn) * sc;
        pz = ug_div_r(pz, nrm, rn) * sc;
      }
      // ((p - lo) / (hi - lo)) * 2 - 1
      const float ux = ug_div_r(px - a.lox, a.ex, a.irx) * 2.f - 1.f;
      const float uy = ug_div_r(py - a.loy, a.ey, a.iry) * 2.f - 1.f;
      const float uz = ug_div_r(pz - a.loz, a.ez, a.irz) * 2.f - 1.f;
      float dens = ug_density_level(bkb, ux, uy, uz, a.X, a.Y, a.Z);
#pragma unroll
      for (int k = 0; k < F; ++k) {
        const float f = (float)(1 << k);
        float sx, cx_, sy, cy_, sz, cz_;
        ug_sincos(f * ux, &sx, &cx_);
        ug_sincos(f * uy, &sy, &cy_);
        ug_sincos(f * uz, &sz, &cz_);
        dens += ug_density_level(bkb + (size_t)(2 * k + 1) * lvl_bytes, sx, sy, sz, a.X, a.Y, a.Z);
        dens += ug_density_level(bkb + (size_t)(2 * k + 2) * lvl_bytes, cx_, cy_, cz_, a.X, a.Y, a.Z);
      }
      dens = dens / (float)P;
      const float alpha = ug_alpha(dens + a.shift, a.interval);
      if (alpha > a.thres) {
        w = T * alpha;
        T = (float)((double)T * (1. - (double)alpha));
        if (w > a.thres) {
          surv = true;
          dsum += w * s_table[j];
        }
        if ((double)T < 1e-3) done = true;
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
  }
  if (lane == 0) ws.count[tile] = nsurv;
}

// ----------------------------------------------------------------------------------------------
// rgbnet packing for the transposed MFMA chain
// packed (floats): A1 [KL][64][4] | A2 [64][64][4] | bias1 [2][64] | bias2 [2][64] | W3 [2][64][4] | b3 [4]
// ----------------------------------------------------------------------------------------------
__host__ __device__ static inline int ug_feat_of(int o, int r, int h) { return 32 * o + (r & 3) + 8 * (r >> 2) + 4 * h; }

struct ug_mlp_layout { int KL, offA1, offA2, offB1, offB2, offW3, offb3, total; };
__host__ __device__ static inline ug_mlp_layout ug_mlp_lay(int C, int n_emb) {
  const int CH = (C + 1) / 2;
  ug_mlp_layout L;
  L.KL = (2 * CH + n_emb + 1) / 2;
  L.offA1 = 0;
  L.offA2 = L.offA1 + L.KL * 256;
  L.offB1 = L.offA2 + 64 * 256;
  L.offB2 = L.offB1 + 128;
  L.offW3 = L.offB2 + 128;
  L.offb3 = L.offW3 + 512;
  L.total = L.offb3 + 4;
  return L;
}

// original rgbnet input column of (step s, half h); -1 = zero padding
__host__ __device__ static inline int ug_in_col(int s, int h, int C, int n_emb, int KL) {
  const int CH = (C + 1) / 2;
  if (s < CH) {
    const int ch = h * CH + s;
    return ch < C ? ch : -1;
  }
  const int e = h * (KL - CH) + (s - CH);
  return e < n_emb ? C + e : -1;
}

__global__ void k_pack_mlp(const float *__restrict__ w0, const float *__restrict__ b0,
                           const float *__restrict__ w1, const float *__restrict__ b1,
                           const float *__restrict__ w2, const float *__restrict__ b2, int C, int n_emb,
                           float *__restrict__ out) {
  const ug_mlp_layout L = ug_mlp_lay(C, n_emb);
  const int mlp_in = C + n_emb;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L.total; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < L.offA2) {                       // A1[s][lane][o] = W0[32o + (lane&31)][col(s, lane>>5)]
      const int o = i & 3, lane = (i >> 2) & 63, s = i >> 8;
      const int col = ug_in_col(s, lane >> 5, C, n_emb, L.KL);
      if (col >= 0) v = w0[(32 * o + (lane & 31)) * mlp_in + col];
    } else if (i < L.offB1) {                // A2[(o',r)][lane][o] = W1[32o + (lane&31)][feat(o',r,lane>>5)]
      const int q = i - L.offA2;
      const int o = q & 3, lane = (q >> 2) & 63, st = q >> 8;
      v = w1[(32 * o + (lane & 31)) * 128 + ug_feat_of(st >> 4, st & 15, lane >> 5)];
    } else if (i < L.offB2) {                // bias1[h][o*16+r]
      const int q = i - L.offB1;
      v = b0[ug_feat_of((q & 63) >> 4, q & 15, q >> 6)];
    } else if (i < L.offW3) {
      const int q = i - L.offB2;
      v = b1[ug_feat_of((q & 63) >> 4, q & 15, q >> 6)];
    } else if (i < L.offb3) {                // W3[h][o*16+r][c]
      const int q = i - L.offW3;
      const int c = q & 3, st = (q >> 2) & 63, h = q >> 8;
      if (c < 3) v = w2[c * 128 + ug_feat_of(st >> 4, st & 15, h)];
    } else {
      const int c = i - L.offb3;
      if (c < 3) v = b2[c];
    }
    out[i] = v;
  }
}

// ----------------------------------------------------------------------------------------------
// shade
// ----------------------------------------------------------------------------------------------
struct ug_shade_args {
  int64_t n_rays;
  int32_t X, Y, Z;
  float lox, loy, loz, hix, hiy, hiz;
  float ex, ey, ez, irx, iry, irz;  // extent hi-lo and RN(1/extent)
};

// one k0 level for one survivor half: in-range cell set-up + one contiguous (8*CH)-float half-brick,
// trilinear per channel in grid_sample's corner order; adds into feat[] (first = level 0 initialises).
template <int CH>
__device__ __forceinline__ void ug_k0_level(const float *__restrict__ k0b, int h, int64_t level_base, float cx,
                                            float cy, float cz, int X, int Y, int Z, bool first, float (&feat)[CH]) {
  const ug_axis_fast ax = ug_axis_inrange(cx, X), ay = ug_axis_inrange(cy, Y), az = ug_axis_inrange(cz, Z);
  const int64_t rec = level_base + ((int64_t)ax.cell * (Y - 1) + ay.cell) * (Z - 1) + az.cell;
  const float *rec_p = k0b + (rec * 2 + h) * (8 * CH);
  float v[8 * CH];
  if constexpr ((8 * CH) % 4 == 0) {
    const float4 *r4 = (const float4 *)rec_p;
#pragma unroll
    for (int q = 0; q < 2 * CH; ++q) {
      const float4 t = r4[q];
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8 * CH; ++q) v[q] = rec_p[q];
  }
  const float w00 = az.wlo * ay.wlo, w01 = az.whi * ay.wlo, w10 = az.wlo * ay.whi, w11 = az.whi * ay.whi;
  const float w[8] = {w00 * ax.wlo, w01 * ax.wlo, w10 * ax.wlo, w11 * ax.wlo,
                      w00 * ax.whi, w01 * ax.whi, w10 * ax.whi, w11 * ax.whi};
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    float acc = v[ch] * w[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) acc += v[c * CH + ch] * w[c];
    feat[ch] = first ? acc : feat[ch] + acc;
  }
}

// k0 half-brick gather for one survivor: CH channels of half h, mean over P = 1+2F levels
// (level order u, sin u, cos u, sin 2u, cos 2u, ... -- FourierGrid_grid.py:70)
template <int F, int CH>
__device__ __forceinline__ void ug_k0_gather(const float *__restrict__ k0b, int h, float px, float py, float pz,
                                             const ug_shade_args &a, float (&feat)[CH]) {
  constexpr int P = 2 * F + 1;
  const float ux = ug_div_r(px - a.lox, a.ex, a.irx) * 2.f - 1.f;
  const float uy = ug_div_r(py - a.loy, a.ey, a.iry) * 2.f - 1.f;
  const float uz = ug_div_r(pz - a.loz, a.ez, a.irz) * 2.f - 1.f;
  const int64_t cells = (int64_t)(a.X - 1) * (a.Y - 1) * (a.Z - 1);
  ug_k0_level<CH>(k0b, h, 0, ux, uy, uz, a.X, a.Y, a.Z, true, feat);
#pragma unroll 1
  for (int k = 0; k < F; ++k) {
    const float f = (float)(1 << k);
    float sx, cx_, sy, cy_, sz, cz_;
    ug_sincos(f * ux, &sx, &cx_);
    ug_sincos(f * uy, &sy, &cy_);
    ug_sincos(f * uz, &sz, &cz_);
    ug_k0_level<CH>(k0b, h, (int64_t)(2 * k + 1) * cells, sx, sy, sz, a.X, a.Y, a.Z, false, feat);
    ug_k0_level<CH>(k0b, h, (int64_t)(2 * k + 2) * cells, cx_, cy_, cz_, a.X, a.Y, a.Z, false, feat);
  }
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) feat[ch] = feat[ch] / (float)P;
}

__device__ __forceinline__ float ug_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// C = 2*CH or 2*CH-1 k0 channels, PE view-direction frequencies; rgbnet 128 wide, 3 layers.
template <int F, int C, int PE, int NW>
__global__ void __launch_bounds__(NW * 64, NW / 4)
k_shade_mlp(ug_shade_args a, const float *__restrict__ viewdirs, const float *__restrict__ k0b,
            const float *__restrict__ mlp, ug_ws_view ws, float *__restrict__ rgb_marched,
            int32_t *__restrict__ tile_counter) {
  constexpr int CH = (C + 1) / 2;
  constexpr int NEMB = 3 + 6 * PE;
  constexpr int KL = (2 * CH + NEMB + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const ug_mlp_layout ML = ug_mlp_lay(C, NEMB);
  for (int i = threadIdx.x; i < ML.total; i += blockDim.x) lds[i] = mlp[i];
  __syncthreads();
  const float4 *A1 = (const float4 *)(lds + ML.offA1);
  const float4 *A2 = (const float4 *)(lds + ML.offA2);
  const float *B1 = lds + ML.offB1, *B2 = lds + ML.offB2;
  const float4 *W3 = (const float4 *)(lds + ML.offW3);
  const float *b3 = lds + ML.offb3;

  const int lane = ug_lane();
  const int h = lane >> 5, sv = lane & 31;

  // dynamic tile scheduling with XCD affinity: the tile range is cut into 8 contiguous eighths (the same
  // eighths k_march assigned to the XCDs), one atomic counter each; a workgroup (XCD = blockIdx % 8) drains
  // its own eighth first, then steals from the others.  Placement only affects speed, never results.
  const int64_t per = (ws.n_tiles + 7) / 8;
  const int home = blockIdx.x & 7;
  int victim = 0;
  for (;;) {
    int64_t tile = -1;
    while (victim < 8) {
      const int q = (home + victim) & 7;
      int t = 0;
      if (lane == 0) t = atomicAdd(tile_counter + q, 1);
      t = __builtin_amdgcn_readfirstlane(t);
      const int64_t cand = (int64_t)q * per + t;
      if (t < per && cand < ws.n_tiles) { tile = cand; break; }
      ++victim;
    }
    if (tile < 0) break;
    const int count = ws.count[tile];
    const float4 *__restrict__ ent = ws.ent + tile * ws.cap;
    const uint8_t *__restrict__ slot = ws.slot + tile * ws.cap;
    float accr = 0.f, accg = 0.f, accb = 0.f;  // lane = ray slot of this tile

    for (int base = 0; base < count; base += 32) {
      const int e = base + sv;
      const bool ok = e < count;
      float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
      int sl = 0;
      if (ok) { en = ent[e]; sl = slot[e]; }
      // ---- layer-1 inputs of this lane: half of k0 + half of the view-direction embedding
      float x[KL];
      {
        float feat[CH];
        ug_k0_gather<F, CH>(k0b, h, en.x, en.y, en.z, a, feat);
#pragma unroll
        for (int s = 0; s < CH; ++s) x[s] = (h * CH + s < C) ? feat[s] : 0.f;
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
      // ---- layer 1: acc1[o] = W0 . x  (A from LDS, B = x registers)
      f32x16 acc1[4], acc2[4];
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[o][r] = B1[h * 64 + o * 16 + r];
#pragma unroll
      for (int s = 0; s < KL; ++s) {
        const float4 wa = A1[s * 64 + lane];
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
          acc1[o][r] = fmaxf(acc1[o][r], 0.f);
          acc2[o][r] = B2[h * 64 + o * 16 + r];
        }
      // ---- layer 2: the accumulators of layer 1 ARE the B operands (k = lane>>5 picks feature +0/+4)
#pragma unroll
      for (int st = 0; st < 64; ++st) {
        const float4 wa = A2[st * 64 + lane];
        const float xb = acc1[st >> 4][st & 15];
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.x, xb, acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.y, xb, acc2[1], 0, 0, 0);
        acc2[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.z, xb, acc2[2], 0, 0, 0);
        acc2[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.w, xb, acc2[3], 0, 0, 0);
        if ((st & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
      // ---- layer 3 (3 outputs) on the VALU: each lane of the pair reduces its 64 features
      float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
      for (int st = 0; st < 64; ++st) {
        const float hv = fmaxf(acc2[st >> 4][st & 15], 0.f);
        const float4 w3 = W3[h * 64 + st];
        l0 = fmaf(w3.x, hv, l0);
        l1 = fmaf(w3.y, hv, l1);
        l2 = fmaf(w3.z, hv, l2);
      }
      l0 = (l0 + __shfl_xor(l0, 32)) + b3[0];
      l1 = (l1 + __shfl_xor(l1, 32)) + b3[1];
      l2 = (l2 + __shfl_xor(l2, 32)) + b3[2];
      // weights.unsqueeze(-1) * rgb, then a per-ray sum in sample order (segment_coo semantics)
      const float pr = en.w * ug_sigmoid(l0), pg = en.w * ug_sigmoid(l1), pb = en.w * ug_sigmoid(l2);
      const int cnt = (count - base) < 32 ? (count - base) : 32;
      for (int k = 0; k < cnt; ++k) {
        const int sk = __builtin_amdgcn_readlane(sl, k);
        const float r_ = ug_readlane_f(pr, k), g_ = ug_readlane_f(pg, k), b_ = ug_readlane_f(pb, k);
        if (lane == sk) { accr += r_; accg += g_; accb += b_; }
      }
    }
    const int64_t ray = tile * UG_WAVE + lane;
    if (ray < a.n_rays) {
      rgb_marched[3 * ray] = accr;
      rgb_marched[3 * ray + 1] = accg;
      rgb_marched[3 * ray + 2] = accb;
    }
  }
}

// rgbnet == None: rgb = sigmoid(k0), k0 is a single-level 3-channel grid (bricks [8][4], ch 3 = 0)
__global__ void __launch_bounds__(256)
k_shade_direct(ug_shade_args a, const float *__restrict__ k0b, ug_ws_view ws,
               float *__restrict__ rgb_marched) {
  const int lane = ug_lane();
  const int64_t tile = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (tile >= ws.n_tiles) return;
  const int count = ws.count[tile];
  const float4 *__restrict__ ent = ws.ent + tile * ws.cap;
  const uint8_t *__restrict__ slot = ws.slot + tile * ws.cap;
  float accr = 0.f, accg = 0.f, accb = 0.f;
  for (int base = 0; base < count; base += UG_WAVE) {
    const int e = base + lane;
    const bool ok = e < count;
    float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
    int sl = 0;
    if (ok) { en = ent[e]; sl = slot[e]; }
    const float ux = ug_div_r(en.x - a.lox, a.ex, a.irx) * 2.f - 1.f;
    const float uy = ug_div_r(en.y - a.loy, a.ey, a.iry) * 2.f - 1.f;
    const float uz = ug_div_r(en.z - a.loz, a.ez, a.irz) * 2.f - 1.f;
    const ug_cellw cw = ug_cell_setup(ux, uy, uz, a.X, a.Y, a.Z, 0);
    const float4 *rec = (const float4 *)(k0b + cw.rec * 32);
    float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 v = rec[c];
      if (c == 0) { f0 = v.x * cw.w[0]; f1 = v.y * cw.w[0]; f2 = v.z * cw.w[0]; }
      else { f0 += v.x * cw.w[c]; f1 += v.y * cw.w[c]; f2 += v.z * cw.w[c]; }
    }
    const float pr = en.w * ug_sigmoid(f0), pg = en.w * ug_sigmoid(f1), pb = en.w * ug_sigmoid(f2);
    const int cnt = (count - base) < UG_WAVE ? (count - base) : UG_WAVE;
    for (int k = 0; k < cnt; ++k) {
      const int sk = __builtin_amdgcn_readlane(sl, k);
      const float r_ = ug_readlane_f(pr, k), g_ = ug_readlane_f(pg, k), b_ = ug_readlane_f(pb, k);
      if (lane == sk) { accr += r_; accg += g_; accb += b_; }
    }
  }
  const int64_t ray = tile * UG_WAVE + lane;
  if (ray < a.n_rays) {
    rgb_marched[3 * ray] = accr;
    rgb_marched[3 * ray + 1] = accg;
    rgb_marched[3 * ray + 2] = accb;
  }
}

__global__ void k_ws_stats(const int32_t *__restrict__ count, int64_t n_tiles, int64_t *__restrict__ out) {
  int64_t s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_tiles; i += (int64_t)gridDim.x * blockDim.x)
    s += count[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (ug_lane() == 0 && s) atomicAdd((unsigned long long *)out, (unsigned long long)s);
}

// ----------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------
#define ST(s) ((hipStream_t)(s))

extern "C" int ugrid_grid_query(const float *grid, int P, int C, int X, int Y, int Z, const float *xyz,
                                const float *xyz_min, const float *xyz_max, int freq_num, int64_t n,
                                float *out, ugrid_stream_t s) {
  if (n <= 0) return 0;
  if (P != (freq_num > 0 ? 2 * freq_num + 1 : 1)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_grid_query, dim3(ug_blocks(n, 256)), dim3(256), 0, ST(s), grid, P, C, X, Y, Z, xyz,
                     xyz_min, xyz_max, freq_num, n, out);
  UG_LAUNCH_CHECK();
  return 0;
}

static inline int ug_brick_ch(int C, int *H) {
  // density (C==1): 1 half x 1 channel; rgbnet-less k0 (C==3 handled by caller via halves=1, CH=4);
  // feature grids: 2 halves x ceil(C/2)
  if (C == 1) { *H = 1; return 1; }
  *H = 2;
  return (C + 1) / 2;
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
  hipLaunchKernelGGL(k_pack_bricks, dim3((unsigned)blocks), dim3(256), 0, ST(s), grid, P, C, X, Y, Z, H, CH,
                     bricks, total);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t ugrid_mlp_packed_bytes(int32_t k0_channels, int32_t viewbase_pe) {
  return (int64_t)sizeof(float) * ug_mlp_lay(k0_channels, 3 + 6 * viewbase_pe).total;
}

extern "C" int ugrid_pack_mlp(const float *w0, const float *b0, const float *w1, const float *b1,
                              const float *w2, const float *b2, int32_t k0_channels, int32_t viewbase_pe,
                              int32_t width, float *packed, ugrid_stream_t s) {
  if (width != 128) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pack_mlp, dim3(64), dim3(256), 0, ST(s), w0, b0, w1, b1, w2, b2, (int)k0_channels,
                     3 + 6 * (int)viewbase_pe, packed);
  UG_LAUNCH_CHECK();
  return 0;
}

template <int F>
static int ug_march_launch(const ugrid_render_params *p, const ug_march_args &a, const float *rays_o,
                           const float *rays_d, const float *t_table, const float *s_table,
                           const float *bricks, float *alphainv_last, float *depth, ug_ws_view ws,
                           hipStream_t st) {
  const int64_t nblocks = (ws.n_tiles + 3) / 4;
  const int64_t grid = ((nblocks + 7) / 8) * 8;  // room for the XCD remap
  if (p->norm_l2)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_march<F, true>), dim3((unsigned)grid), dim3(256), 0, st, a, rays_o,
                       rays_d, t_table, s_table, bricks, alphainv_last, depth, ws, nblocks);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_march<F, false>), dim3((unsigned)grid), dim3(256), 0, st, a, rays_o,
                       rays_d, t_table, s_table, bricks, alphainv_last, depth, ws, nblocks);
  UG_LAUNCH_CHECK();
  return 0;
}

extern "C" int ugrid_render_march(const ugrid_render_params *p, const float *rays_o, const float *rays_d,
                                  const float *t_table, const float *s_table, const float *density_bricks,
                                  float *alphainv_last, float *depth, void *ws_mem, ugrid_stream_t s) {
  if (p->n_rays <= 0) return 0;
  if (p->n_samples <= 0 || p->grid_x < 2 || p->grid_y < 2 || p->grid_z < 2) return (int)hipErrorInvalidValue;
  ug_ws_view ws = ug_ws_make(ws_mem, p->n_rays, p->n_samples);
  ug_march_args a;
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
  switch (p->freq_num) {
    case 1: return ug_march_launch<1>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    case 2: return ug_march_launch<2>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    case 3: return ug_march_launch<3>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    case 4: return ug_march_launch<4>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    case 5: return ug_march_launch<5>(p, a, rays_o, rays_d, t_table, s_table, density_bricks, alphainv_last, depth, ws, ST(s));
    default: return (int)hipErrorInvalidValue;
  }
}

static int g_shade_waves = 12;  // waves per shade workgroup: 8 (2/SIMD, no spills) or 12 (3/SIMD)

extern "C" int ugrid_tune(const char *key, int value) {
  if (!key) return (int)hipErrorInvalidValue;
  if (!strcmp(key, "shade_waves") && (value == 8 || value == 12)) { g_shade_waves = value; return 0; }
  return (int)hipErrorInvalidValue;
}

template <int F, int C, int PE, int NW>
static int ug_shade_launch_nw(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                              ug_ws_view ws, float *rgb, int32_t *counter, hipStream_t st) {
  const int lds_bytes = (int)sizeof(float) * ug_mlp_lay(C, 3 + 6 * PE).total;
  static bool attr_set = false;
  if (!attr_set) {
    UG_HIP(hipFuncSetAttribute((const void *)k_shade_mlp<F, C, PE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_set = true;
  }
  UG_HIP(hipMemsetAsync(counter, 0, 8 * sizeof(int32_t), st));
  // persistent: one workgroup per CU (LDS holds the 89 KB packed rgbnet)
  int64_t wgs = (ws.n_tiles + NW - 1) / NW;
  if (wgs > 256) wgs = 256;
  wgs = (wgs + 7) / 8 * 8;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shade_mlp<F, C, PE, NW>), dim3((unsigned)wgs), dim3(NW * 64), lds_bytes, st, a,
                     viewdirs, k0b, mlp, ws, rgb, counter);
  UG_LAUNCH_CHECK();
  return 0;
}

template <int F, int C, int PE>
static int ug_shade_launch(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                           ug_ws_view ws, float *rgb, int32_t *counter, hipStream_t st) {
  if (g_shade_waves == 8) return ug_shade_launch_nw<F, C, PE, 8>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
  return ug_shade_launch_nw<F, C, PE, 12>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
}

extern "C" int ugrid_render_shade(const ugrid_render_params *p, const float *viewdirs, const float *k0_bricks,
                                  const float *mlp_packed, void *ws_mem, float *rgb_marched, ugrid_stream_t s) {
  if (p->n_rays <= 0) return 0;
  ug_ws_view ws = ug_ws_make(ws_mem, p->n_rays, p->n_samples);
  ug_shade_args a;
  a.n_rays = p->n_rays; a.X = p->grid_x; a.Y = p->grid_y; a.Z = p->grid_z;
  a.lox = p->xyz_min[0]; a.loy = p->xyz_min[1]; a.loz = p->xyz_min[2];
  a.hix = p->xyz_max[0]; a.hiy = p->xyz_max[1]; a.hiz = p->xyz_max[2];
  a.ex = a.hix - a.lox; a.ey = a.hiy - a.loy; a.ez = a.hiz - a.loz;
  a.irx = 1.0f / a.ex; a.iry = 1.0f / a.ey; a.irz = 1.0f / a.ez;
  if (p->mlp_in == 0) {
    if (p->k0_channels != 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_shade_direct, dim3(ug_blocks(ws.n_tiles * UG_WAVE, 256)), dim3(256), 0, ST(s), a,
                       k0_bricks, ws, rgb_marched);
    UG_LAUNCH_CHECK();
    return 0;
  }
  if (p->mlp_width != 128 || p->mlp_in != p->k0_channels + 3 + 6 * p->viewbase_pe) return (int)hipErrorInvalidValue;
  int32_t *counter = (int32_t *)ws_mem;  // first 256 B of the work list
#define UG_SHADE_CASE(F_, C_, PE_)                                                          \
  if (p->freq_num == F_ && p->k0_channels == C_ && p->viewbase_pe == PE_)                   \
    return ug_shade_launch<F_, C_, PE_>(a, viewdirs, k0_bricks, mlp_packed, ws, rgb_marched, counter, ST(s));
  UG_SHADE_CASE(3, 12, 4)  // Mip-NeRF-360 *_single.py  (configs/default.py:104-124)
  UG_SHADE_CASE(4, 12, 4)  // tankstemple_unbounded/truck_single.py:105
  UG_SHADE_CASE(2, 3, 2)   // waymo-style rgbnet_dim=3, viewbase_pe=2 (configs/waymo/waymo_no_block.py:144-149)
  UG_SHADE_CASE(3, 3, 2)
  UG_SHADE_CASE(1, 12, 4)
#undef UG_SHADE_CASE
  return (int)hipErrorNotSupported;
}

extern "C" int ugrid_render_stats(void *ws_mem, int64_t n_rays, int32_t S, int64_t *d_stats, ugrid_stream_t s) {
  ug_ws_view ws = ug_ws_make(ws_mem, n_rays, S);
  UG_HIP(hipMemsetAsync(d_stats, 0, sizeof(int64_t), ST(s)));
  hipLaunchKernelGGL(k_ws_stats, dim3(64), dim3(256), 0, ST(s), ws.count, ws.n_tiles, d_stats);
  UG_LAUNCH_CHECK();
  return 0;
}
