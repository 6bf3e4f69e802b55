// EXPERIMENTAL k0-gather variants (symbols ugx_*; not part of the C ABI in include/ugrid_hip.h).  Timed on the real S1
// work list by tools/gpu_gather_variants.py to decide the brick layout / lane mapping of the shade kernel:
//   pair   (id 0)  the layout the shade kernel used in round 1: lanes l / l+32 own a survivor, each reads its 192-byte
//                  half-brick with 12 dwordx4 loads -- 64 different 16-byte pieces per load instruction
//   quad   (id 1+) 4 ADJACENT lanes own a survivor; brick = [cell][q 0..5][g 0..3][4 floats]: load q of the quad reads
//                  64 contiguous, 64-byte aligned bytes; lane g ends up with all 8 coefficients of channels 3g..3g+2
//   vertex         canonical values re-laid as [P][X][Y][Z][C] (48 B per vertex, 8x smaller than bricks): lane g reads
//                  3 channels of each of the 8 corners (dwordx3), weighted corner sum in grid_sample's own order
// All variants write the survivors' mean-over-levels k0 features to ws.feat ([entry][12]) like k_shade_gather.
#include "ugrid_render.h"

// ---- packing --------------------------------------------------------------------------------------------------
__global__ void k_pack_quad(const float *__restrict__ grid, int P, int C, int X, int Y, int Z, float *__restrict__ out,
                            int64_t total);   // ugrid_march.hip
extern "C" int ug_pack_bricks_pair(const float *grid, int P, int C, int X, int Y, int Z, float *bricks, ugrid_stream_t s);

__global__ void k_pack_vertex(const float *__restrict__ grid, int P, int C, int X, int Y, int Z, float *__restrict__ out,
                              int64_t total) {
  const int64_t vol = (int64_t)X * Y * Z;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(o % C);
    const int64_t r = o / C;           // l * vol + voxel
    const int64_t l = r / vol, vox = r - l * vol;
    out[o] = grid[(l * C + ch) * vol + vox];
  }
}

extern "C" int64_t ugx_pack_bytes(int P, int C, int X, int Y, int Z, int layout) {
  if (layout <= 1) return (int64_t)P * (X - 1) * (Y - 1) * (Z - 1) * 96 * 4;
  return (int64_t)P * X * Y * Z * C * 4;
}

extern "C" int ugx_pack(const float *grid, int P, int C, int X, int Y, int Z, int layout, float *out, ugrid_stream_t s) {
  if (C != 12) return (int)hipErrorInvalidValue;
  const int64_t total = ugx_pack_bytes(P, C, X, Y, Z, layout) / 4;
  if (layout == 0) return ug_pack_bricks_pair(grid, P, C, X, Y, Z, out, s);
  if (layout == 1)
    hipLaunchKernelGGL(k_pack_quad, dim3(256 * 64), dim3(256), 0, (hipStream_t)s, grid, P, C, X, Y, Z, out, total);
  else
    hipLaunchKernelGGL(k_pack_vertex, dim3(256 * 64), dim3(256), 0, (hipStream_t)s, grid, P, C, X, Y, Z, out, total);
  UG_LAUNCH_CHECK();
  return 0;
}

// dwordx3 flavour of the explicit loads (ugrid_render.h) for the vertex layout
typedef float ug_f3v __attribute__((ext_vector_type(3)));
template <int IMM>
__device__ __forceinline__ ug_f3v ug_gload3(unsigned voff, const float *sbase) {
  ug_f3v r;
  asm volatile("global_load_dwordx3 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void ug_vmwait8(ug_f3v (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "n"(N));
}

// ---- quad helpers ------------------------------------------------------------------------------------------------
struct ug_qcell { unsigned cell; float tx, ty, tz; };

// every lane computes the whole cell set-up of one level (4x redundant inside a quad)
__device__ __forceinline__ ug_qcell ug_qcell_full(float cx, float cy, float cz, int X, int Y, int Z) {
  const ug_axis_fast ax = ug_axis_inrange(cx, X), ay = ug_axis_inrange(cy, Y), az = ug_axis_inrange(cz, Z);
  const unsigned row = (unsigned)fmaf(ax.cellf, (float)(Y - 1), ay.cellf);
  ug_qcell c;
  c.cell = __umul24(row, (unsigned)(Z - 1)) + (unsigned)az.cell;
  c.tx = ax.whi; c.ty = ay.whi; c.tz = az.whi;
  return c;
}

// lane g of the quad has set up ITS axis (g = 0: x, 1: y, 2 / 3: z); combine through DPP broadcasts
__device__ __forceinline__ ug_qcell ug_qcell_shared(const ug_axis_fast &mine, int Y, int Z) {
  const float cxf = ug_quad_bcast<0>(mine.cellf), cyf = ug_quad_bcast<1>(mine.cellf), czf = ug_quad_bcast<2>(mine.cellf);
  const unsigned row = (unsigned)fmaf(cxf, (float)(Y - 1), cyf);
  ug_qcell c;
  c.cell = __umul24(row, (unsigned)(Z - 1)) + (unsigned)czf;
  c.tx = ug_quad_bcast<0>(mine.whi); c.ty = ug_quad_bcast<1>(mine.whi); c.tz = ug_quad_bcast<2>(mine.whi);
  return c;
}

// MODE 0: redundant set-up; 1: set-up shared across the quad; 2: as 1 but every quad reads the cell of quad 0
// (diagnostic: removes the address divergence between quads, keeps the instruction stream)
template <int F, int NBL, int WPS, int MODE>
__global__ void __launch_bounds__(256, WPS)
k_gather_quad(ug_shade_args a, const float *__restrict__ qb, ug_ws_view ws, float *__restrict__ featbuf, int64_t nblocks) {
  constexpr int P = 2 * F + 1;
  const int64_t blk = ug_xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t tile = blk * 4 + (threadIdx.x >> 6);
  if (tile >= ws.n_tiles) return;
  const int lane = ug_lane(), s = lane >> 2, g = lane & 3;
  const int count = ws.count[tile];
  const float4 *__restrict__ ent = ws.ent + tile * ws.cap;
  float *__restrict__ fo = featbuf + tile * ws.cap * UG_FEAT_STRIDE;
  const int64_t cells = (int64_t)(a.X - 1) * (a.Y - 1) * (a.Z - 1);
  // per-lane axis constants (shared set-up): lane g works on axis min(g, 2)
  const float lo_g = g == 0 ? a.lox : (g == 1 ? a.loy : a.loz);
  const float ex_g = g == 0 ? a.ex : (g == 1 ? a.ey : a.ez);
  const float ir_g = g == 0 ? a.irx : (g == 1 ? a.iry : a.irz);
  const int n_g = g == 0 ? a.X : (g == 1 ? a.Y : a.Z);
  for (int base = 0; base < count; base += 16) {
    const int e = base + s;
    const bool ok = e < count;
    float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) en = ent[e];
    ug_qcell c[P];
    if constexpr (MODE == 0) {
      const float ux = ug_div_r(en.x - a.lox, a.ex, a.irx) * 2.f - 1.f;
      const float uy = ug_div_r(en.y - a.loy, a.ey, a.iry) * 2.f - 1.f;
      const float uz = ug_div_r(en.z - a.loz, a.ez, a.irz) * 2.f - 1.f;
      c[0] = ug_qcell_full(ux, uy, uz, a.X, a.Y, a.Z);
#pragma unroll
      for (int k = 0; k < F; ++k) {
        const float f = (float)(1 << k);
        float sx, cx_, sy, cy_, sz, cz_;
        ug_sincos(f * ux, &sx, &cx_);
        ug_sincos(f * uy, &sy, &cy_);
        ug_sincos(f * uz, &sz, &cz_);
        c[2 * k + 1] = ug_qcell_full(sx, sy, sz, a.X, a.Y, a.Z);
        c[2 * k + 2] = ug_qcell_full(cx_, cy_, cz_, a.X, a.Y, a.Z);
      }
    } else {
      const float p_g = g == 0 ? en.x : (g == 1 ? en.y : en.z);
      const float u = ug_div_r(p_g - lo_g, ex_g, ir_g) * 2.f - 1.f;
      c[0] = ug_qcell_shared(ug_axis_inrange(u, n_g), a.Y, a.Z);
#pragma unroll
      for (int k = 0; k < F; ++k) {
        float sn, cs;
        ug_sincos((float)(1 << k) * u, &sn, &cs);
        c[2 * k + 1] = ug_qcell_shared(ug_axis_inrange(sn, n_g), a.Y, a.Z);
        c[2 * k + 2] = ug_qcell_shared(ug_axis_inrange(cs, n_g), a.Y, a.Z);
      }
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int l = 0; l < P; ++l) c[l].cell = (unsigned)__builtin_amdgcn_readfirstlane((int)c[l].cell);
    }
    float feat[3] = {0.f, 0.f, 0.f};
    // software pipeline over the levels, NBL levels (x 6 loads) in flight; byte offsets stay below 4 GiB per level
    ug_f4 v[NBL][6];
    unsigned off[P];
#pragma unroll
    for (int l = 0; l < P; ++l) off[l] = c[l].cell * 384u + (unsigned)g * 16u;
#define UGX_ISSUE(l_)                                                                   \
    {                                                                                   \
      const float *lb = qb + (int64_t)(l_) * cells * 96;                                \
      v[(l_) % NBL][0] = ug_gload4<0>(off[l_], lb);   v[(l_) % NBL][1] = ug_gload4<64>(off[l_], lb);   \
      v[(l_) % NBL][2] = ug_gload4<128>(off[l_], lb); v[(l_) % NBL][3] = ug_gload4<192>(off[l_], lb);  \
      v[(l_) % NBL][4] = ug_gload4<256>(off[l_], lb); v[(l_) % NBL][5] = ug_gload4<320>(off[l_], lb);  \
    }
#pragma unroll
    for (int l = 0; l < NBL && l < P; ++l) UGX_ISSUE(l)
#pragma unroll
    for (int l = 0; l < P; ++l) {
      constexpr int kAll = 0;
      (void)kAll;
      const int after = (P - 1 - l) < (NBL - 1) ? (P - 1 - l) : (NBL - 1);   // levels issued after level l
      if (after == 0) ug_vmwait6<0>(v[l % NBL]);
      else if (after == 1) ug_vmwait6<6>(v[l % NBL]);
      else if (after == 2) ug_vmwait6<12>(v[l % NBL]);
      else if (after == 3) ug_vmwait6<18>(v[l % NBL]);
      else if (after == 4) ug_vmwait6<24>(v[l % NBL]);
      else if (after == 5) ug_vmwait6<30>(v[l % NBL]);
      else ug_vmwait6<36>(v[l % NBL]);
      ug_quad_poly(v[l % NBL], c[l].tx, c[l].ty, c[l].tz, l == 0, feat);
      // the per-level math must stay here: LLVM's Sink pass moves it into the `if (ok)` block below (its only user),
      // which keeps all 7 levels' loads alive (spills); a volatile asm use pins it
      asm volatile("" :: "v"(feat[0]), "v"(feat[1]), "v"(feat[2]));
      __builtin_amdgcn_sched_barrier(0);
      if (l + NBL < P) UGX_ISSUE(l + NBL)
    }
#undef UGX_ISSUE
    if (ok) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        fo[(int64_t)e * UG_FEAT_STRIDE + 3 * g + ch] = ug_div_r(feat[ch], (float)P, 1.0f / (float)P);
    }
  }
}

// vertex layout [P][X][Y][Z][12]: lane g reads channels 3g..3g+2 of the 8 corners (dwordx3); corner sum in
// grid_sample's accumulation order with its weights (wz*wy)*wx -- separate multiply and add like torch's CPU kernel
struct ug_f3 { float x, y, z; };
template <int F, int NBL, int WPS>
__global__ void __launch_bounds__(256, WPS)
k_gather_vertex(ug_shade_args a, const float *__restrict__ vb, ug_ws_view ws, float *__restrict__ featbuf, int64_t nblocks) {
  constexpr int P = 2 * F + 1;
  const int64_t blk = ug_xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t tile = blk * 4 + (threadIdx.x >> 6);
  if (tile >= ws.n_tiles) return;
  const int lane = ug_lane(), s = lane >> 2, g = lane & 3;
  const int count = ws.count[tile];
  const float4 *__restrict__ ent = ws.ent + tile * ws.cap;
  float *__restrict__ fo = featbuf + tile * ws.cap * UG_FEAT_STRIDE;
  const int64_t vol = (int64_t)a.X * a.Y * a.Z;
  const float lo_g = g == 0 ? a.lox : (g == 1 ? a.loy : a.loz);
  const float ex_g = g == 0 ? a.ex : (g == 1 ? a.ey : a.ez);
  const float ir_g = g == 0 ? a.irx : (g == 1 ? a.iry : a.irz);
  const int n_g = g == 0 ? a.X : (g == 1 ? a.Y : a.Z);
  const unsigned sz = 12u, sy = (unsigned)a.Z * 12u, sx = (unsigned)a.Y * (unsigned)a.Z * 12u;   // corner strides (floats)
  for (int base = 0; base < count; base += 16) {
    const int e = base + s;
    const bool ok = e < count;
    float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) en = ent[e];
    const float p_g = g == 0 ? en.x : (g == 1 ? en.y : en.z);
    const float u = ug_div_r(p_g - lo_g, ex_g, ir_g) * 2.f - 1.f;
    float lc[P];
    lc[0] = u;
#pragma unroll
    for (int k = 0; k < F; ++k) ug_sincos((float)(1 << k) * u, &lc[2 * k + 1], &lc[2 * k + 2]);
    float feat[3] = {0.f, 0.f, 0.f};
    ug_f3v v[NBL][8];
    unsigned off[P];
    float wxl[P], wxh[P], wyl[P], wyh[P], wzl[P], wzh[P];
#pragma unroll
    for (int l = 0; l < P; ++l) {
      const ug_axis_fast mine = ug_axis_inrange(lc[l], n_g);
      const float cxf = ug_quad_bcast<0>(mine.cellf), cyf = ug_quad_bcast<1>(mine.cellf), czf = ug_quad_bcast<2>(mine.cellf);
      wxh[l] = ug_quad_bcast<0>(mine.whi); wyh[l] = ug_quad_bcast<1>(mine.whi); wzh[l] = ug_quad_bcast<2>(mine.whi);
      wxl[l] = ug_quad_bcast<0>(mine.wlo); wyl[l] = ug_quad_bcast<1>(mine.wlo); wzl[l] = ug_quad_bcast<2>(mine.wlo);
      const unsigned row = (unsigned)fmaf(cxf, (float)a.Y, cyf);
      const unsigned vox = __umul24(row, (unsigned)a.Z) + (unsigned)czf;     // < 2^24 voxels per level at G <= 255
      off[l] = vox * 48u + (unsigned)g * 12u;                                 // bytes inside the level (< 4 GiB)
    }
    const unsigned by = (unsigned)a.Z * 48u, bx = (unsigned)a.Y * (unsigned)a.Z * 48u;  // corner strides in bytes
    (void)sz; (void)sy; (void)sx;
#define UGX_ISSUE(l_)                                                                                  \
    {                                                                                                  \
      const float *lb = vb + (int64_t)(l_) * vol * 12;                                                 \
      const unsigned o00 = off[l_], o01 = off[l_] + by, o10 = off[l_] + bx, o11 = off[l_] + bx + by;   \
      v[(l_) % NBL][0] = ug_gload3<0>(o00, lb); v[(l_) % NBL][1] = ug_gload3<48>(o00, lb);             \
      v[(l_) % NBL][2] = ug_gload3<0>(o01, lb); v[(l_) % NBL][3] = ug_gload3<48>(o01, lb);             \
      v[(l_) % NBL][4] = ug_gload3<0>(o10, lb); v[(l_) % NBL][5] = ug_gload3<48>(o10, lb);             \
      v[(l_) % NBL][6] = ug_gload3<0>(o11, lb); v[(l_) % NBL][7] = ug_gload3<48>(o11, lb);             \
    }
#pragma unroll
    for (int l = 0; l < NBL && l < P; ++l) UGX_ISSUE(l)
#pragma unroll
    for (int l = 0; l < P; ++l) {
      const int after = (P - 1 - l) < (NBL - 1) ? (P - 1 - l) : (NBL - 1);
      if (after == 0) ug_vmwait8<0>(v[l % NBL]);
      else if (after == 1) ug_vmwait8<8>(v[l % NBL]);
      else if (after == 2) ug_vmwait8<16>(v[l % NBL]);
      else if (after == 3) ug_vmwait8<24>(v[l % NBL]);
      else ug_vmwait8<32>(v[l % NBL]);
      const float zy00 = wzl[l] * wyl[l], zy10 = wzh[l] * wyl[l], zy01 = wzl[l] * wyh[l], zy11 = wzh[l] * wyh[l];
      float w[8];
      w[0] = zy00 * wxl[l]; w[1] = zy10 * wxl[l]; w[2] = zy01 * wxl[l]; w[3] = zy11 * wxl[l];
      w[4] = zy00 * wxh[l]; w[5] = zy10 * wxh[l]; w[6] = zy01 * wxh[l]; w[7] = zy11 * wxh[l];
      const ug_f3v(&vv)[8] = v[l % NBL];
      float f0 = vv[0].x * w[0], f1 = vv[0].y * w[0], f2 = vv[0].z * w[0];
#pragma unroll
      for (int cc = 1; cc < 8; ++cc) {
        f0 = f0 + vv[cc].x * w[cc]; f1 = f1 + vv[cc].y * w[cc]; f2 = f2 + vv[cc].z * w[cc];
      }
      feat[0] = (l == 0) ? f0 : feat[0] + f0;
      feat[1] = (l == 0) ? f1 : feat[1] + f1;
      feat[2] = (l == 0) ? f2 : feat[2] + f2;
      asm volatile("" :: "v"(feat[0]), "v"(feat[1]), "v"(feat[2]));
      __builtin_amdgcn_sched_barrier(0);
      if (l + NBL < P) UGX_ISSUE(l + NBL)
    }
#undef UGX_ISSUE
    if (ok) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        fo[(int64_t)e * UG_FEAT_STRIDE + 3 * g + ch] = ug_div_r(feat[ch], (float)P, 1.0f / (float)P);
    }
  }
}

// the round-1 pair layout, optionally with every lane reading lane 0's cell (diagnostic)
template <int F, int C, bool SAMECELL>
__global__ void __launch_bounds__(256, 4)
k_gather_pair(ug_shade_args a, const float *__restrict__ k0b, ug_ws_view ws, float *__restrict__ featbuf, int64_t nblocks) {
  constexpr int CH = UG_CH(C);
  const int64_t blk = ug_xcd_remap(blockIdx.x, nblocks);
  if (blk >= nblocks) return;
  const int64_t tile = blk * 4 + (threadIdx.x >> 6);
  if (tile >= ws.n_tiles) return;
  const int lane = ug_lane(), h = lane >> 5, sv = lane & 31;
  const int count = ws.count[tile];
  const float4 *__restrict__ ent = ws.ent + tile * ws.cap;
  float *__restrict__ fo = featbuf + tile * ws.cap * UG_FEAT_STRIDE;
  for (int base = 0; base < count; base += 32) {
    const int e = base + sv;
    const bool ok = e < count;
    float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) en = ent[e];
    if (SAMECELL) {
      en.x = ug_readlane_f(en.x, 0); en.y = ug_readlane_f(en.y, 0); en.z = ug_readlane_f(en.z, 0);
    }
    float feat[CH];
    ug_k0_gather<F, CH>(k0b, SAMECELL ? 0 : h, en.x, en.y, en.z, a, feat);
    if (ok) {
#pragma unroll
      for (int s = 0; s < CH; ++s) fo[(int64_t)e * UG_FEAT_STRIDE + h * CH + s] = feat[s];
    }
  }
}

// order-independent exact checksum of the valid feature entries (sum of the fp32 bit patterns) + double sum / sum of squares
__global__ void k_feat_checksum(ug_ws_view ws, const float *__restrict__ featbuf, unsigned long long *__restrict__ out_bits, double *__restrict__ out_sum) {
  unsigned long long bits = 0;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t tile = blockIdx.x; tile < ws.n_tiles; tile += gridDim.x) {
    const int64_t n = (int64_t)ws.count[tile] * UG_FEAT_STRIDE;
    const float *f = featbuf + tile * ws.cap * UG_FEAT_STRIDE;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
      const float v = f[i];
      bits += (unsigned long long)__float_as_uint(v);
      s1 += (double)v; s2 += (double)v * (double)v;
    }
  }
  atomicAdd(out_bits, bits);
  atomicAdd(out_sum, s1);
  atomicAdd(out_sum + 1, s2);
}

extern "C" int64_t ugx_feat_bytes(int64_t n_rays, int32_t S) {
  return ((n_rays + UG_WAVE - 1) / UG_WAVE) * (int64_t)UG_WAVE * S * UG_FEAT_STRIDE * 4;
}

extern "C" int ugx_feat_checksum(void *ws_mem, const float *featbuf, int64_t n_rays, int32_t S, void *out3, ugrid_stream_t s) {
  ug_ws_view ws = ug_ws_make(ws_mem, n_rays, S);
  UG_HIP(hipMemsetAsync(out3, 0, 24, (hipStream_t)s));
  hipLaunchKernelGGL(k_feat_checksum, dim3(2048), dim3(256), 0, (hipStream_t)s, ws, featbuf, (unsigned long long *)out3,
                     (double *)((char *)out3 + 8));
  UG_LAUNCH_CHECK();
  return 0;
}

// variant ids: see tools/gpu_gather_variants.py
extern "C" int ugx_gather(const ugrid_render_params *p, const float *bricks, void *ws_mem, float *featbuf, int variant,
                          ugrid_stream_t s) {
  if (p->freq_num != 3 || p->k0_channels != 12) return (int)hipErrorNotSupported;
  ug_ws_view ws = ug_ws_make(ws_mem, p->n_rays, p->n_samples);
  ug_shade_args a;
  ug_fill_shade_args(p, a);
  const int64_t nblocks = (ws.n_tiles + 3) / 4;
  const dim3 grid((unsigned)(((nblocks + 7) / 8) * 8)), block(256);
  hipStream_t st = (hipStream_t)s;
#define UGX_Q(NBL, WPS, MODE) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_quad<3, NBL, WPS, MODE>), grid, block, 0, st, a, bricks, ws, featbuf, nblocks)
#define UGX_V(NBL, WPS) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_vertex<3, NBL, WPS>), grid, block, 0, st, a, bricks, ws, featbuf, nblocks)
  switch (variant) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_pair<3, 12, false>), grid, block, 0, st, a, bricks, ws, featbuf, nblocks); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather_pair<3, 12, true>), grid, block, 0, st, a, bricks, ws, featbuf, nblocks); break;
    case 10: UGX_Q(2, 4, 0); break;
    case 11: UGX_Q(2, 4, 1); break;
    case 12: UGX_Q(2, 4, 2); break;
    case 13: UGX_Q(4, 3, 1); break;
    case 14: UGX_Q(1, 6, 1); break;
    case 15: UGX_Q(7, 2, 1); break;
    case 16: UGX_Q(2, 5, 1); break;
    case 20: UGX_V(1, 4); break;
    case 21: UGX_V(2, 4); break;
    case 22: UGX_V(3, 3); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef UGX_Q
#undef UGX_V
  UG_LAUNCH_CHECK();
  return 0;
}
