// libugrid_hip.so -- shade half of the fused render path (rgbnet on MFMA: fp16x2 / bf16x3 / fp32), the
// single-launch variant and the rgbnet packing kernel.
#include "ugrid_render.h"
#include "ugrid_shade_pc.h"
#ifdef UG_SHADE_PROF
extern "C" int ugx_pc_dbg_set(int bits) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pc_dbg), &bits, sizeof(int)); }
#endif

extern "C" int ug_set_march_waves(int w);  // ugrid_march.hip
extern "C" int ug_set_tv_xcd(int m);        // ugrid_ops.hip
#ifndef UG_PC48_SLOTS
#define UG_PC48_SLOTS 2      // ring slots per consumer of the 4 + 8 geometry
#endif
#ifndef UG_PC57_EXTRA
#define UG_PC57_EXTRA 0      // extra gather items in flight of a producer that feeds ONE consumer (it holds one stream's state less)
#endif
#ifndef UG_PC48_NBL
#define UG_PC48_NBL 4        // gather items in flight per producer of the 4 + 8 geometry (rolling cell set-up: ug_k0_gather_quad_roll)
#endif
#ifndef UG_PC12_NBL
#define UG_PC12_NBL 3        // gather items (x 6 dwordx4) in flight per producer wave of the 12-wave geometry (4 spills: A/B arm only)
#endif
static int g_shade_pc = 2;   // ugrid_tune("shade_pc", 0|1|2): 2 = 12-wave producer / consumer shade kernel where it applies (default),
                             // 1 = its 8-wave form, 0 = the classic one-wave-does-everything kernel -- bit-identical results,
                             // A/B switch for measurements
#ifdef UG_EXPERIMENTS        // csrc/build.sh with UG_EXPERIMENTS=1: the rejected A/B arms, never in the shipped library
static int g_shade_dbg = 0;  // ugrid_tune("shade_dbg", bits) -- WRONG RESULTS by design (see ug_shade_tile16)
static int g_shade16 = 0;    // ugrid_tune("shade16", 0|1): 16x16x32 / 16-wave kernel where it applies
#endif

__global__ void k_pack_mlp(const float *__restrict__ w0, const float *__restrict__ b0,
                           const float *__restrict__ w1, const float *__restrict__ b1,
                           const float *__restrict__ w2, const float *__restrict__ b2, int C, int n_emb,
                           ug_mlp_scales sc, float *__restrict__ out) {
  const ug_mlp_layout L = ug_mlp_lay(C, n_emb);
  const int mlp_in = C + n_emb;
  // fp16x2 image: one thread per fp16 element, same (step, tile, lane, element) walk as the bf16 image
  unsigned short *hx = (unsigned short *)(out + L.hxA1);
  const int n_hx = (L.hxB1 - L.hxA1) * 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_hx; i += gridDim.x * blockDim.x) {
    const int e = i & 7, lane = (i >> 3) & 63, u = i >> 9;      // u = (step*4 + o)*2 + part
    const int part = u & 1, o = (u >> 1) & 3, step = u >> 3;
    float w = 0.f;
    if (step < L.KB1) {
      const int idx = 8 * step + e;
      const int col = idx < L.KL ? ug_in_col(idx, lane >> 5, C, n_emb, L.KL) : -1;
      if (col >= 0) w = w0[(32 * o + (lane & 31)) * mlp_in + col] * sc.sW1;
    } else {
      const int st = step - L.KB1;
      w = w1[(32 * o + (lane & 31)) * 128 + ug_feat_of(st >> 1, 8 * (st & 1) + e, lane >> 5)] * sc.sW2;
    }
    const _Float16 hh = (_Float16)w;
    const _Float16 ll = (_Float16)(w - (float)hh);
    hx[i] = __builtin_bit_cast(unsigned short, part == 0 ? hh : ll);
  }
  if (blockIdx.x == 0 && threadIdx.x < 4)
    out[L.hxS + threadIdx.x] = threadIdx.x == 0 ? sc.sX1 : (threadIdx.x == 1 ? sc.sX2 / (sc.sW1 * sc.sX1) : 0.f);
  // 16x16x32 fp16x2 image (k_shade_mlp16; C = 12, PE = 4): A operands lane (m = lane & 15, jg = lane >> 4) holds
  // W[16 t + m][input of K slot (jg, 8 ks + e)]
  if (C == 12 && n_emb == 27) {
    unsigned short *qx = (unsigned short *)(out + L.qA1);
    const int n_q = (L.qB2 - L.qA1) * 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_q; i += gridDim.x * blockDim.x) {
      const int e = i & 7, lane = (i >> 3) & 63, u = i >> 9;      // u = (ks * 8 + t) * 2 + part, layer 2 after layer 1
      const int part = u & 1, t = (u >> 1) & 7, ksg = u >> 4;
      const int m = lane & 15, jg = lane >> 4, f = 16 * t + m;
      float w = 0.f;
      if (ksg < 2) {
        const int slot = 8 * ksg + e;
        const int col = ug_q16_col(slot, jg);
        if (col >= 0) w = w0[f * mlp_in + col] * sc.sW1;
        else if (col == -2) w = b0[f] * sc.sW1;                 // the constant-one slot carries the layer-1 bias
      } else {
        const int ks = ksg - 2;
        w = w1[f * 128 + ug_q16_feat(ks, jg, e)] * sc.sW2;
      }
      const _Float16 hh = (_Float16)w;
      const _Float16 ll = (_Float16)(w - (float)hh);
      qx[i] = __builtin_bit_cast(unsigned short, part == 0 ? hh : ll);
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 128 + 512 + 8; i += gridDim.x * blockDim.x) {
      float v = 0.f;
      if (i < 128) {                     // bias2 [jg][4 t + r] at the accumulator scale of layer 2
        const int jg = i >> 5, t = (i >> 2) & 7, r = i & 3;
        v = b1[16 * t + 4 * jg + r] * (sc.sW2 * sc.sX2);
      } else if (i < 640) {              // W3 [jg][4 t + r][c]
        const int q = i - 128, c = q & 3, tr = (q >> 2) & 31, jg = q >> 7;
        if (c < 3) v = w2[c * 128 + 16 * (tr >> 2) + 4 * jg + (tr & 3)] / (sc.sW2 * sc.sX2);
      } else if (i < 644) {
        if (i - 640 < 3) v = b2[i - 640];
      } else {
        v = (i == 644) ? sc.sX1 : (i == 645 ? sc.sX2 / (sc.sW1 * sc.sX1) : 0.f);
      }
      out[L.qB2 + i] = v;
    }
  }
  // bf16x3 image: one thread per bf16 element
  unsigned short *bf = (unsigned short *)(out + L.bfA1);
  const int n_bf = (L.bfB1 - L.bfA1) * 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_bf; i += gridDim.x * blockDim.x) {
    const int e = i & 7, lane = (i >> 3) & 63, u = i >> 9;      // u = (step*4 + o)*3 + part
    const int part = u % 3, o = (u / 3) & 3, step = u / 12;
    float w = 0.f;
    if (step < L.KB1) {                                        // layer 1: inputs 8*step+e of half (lane>>5)
      const int idx = 8 * step + e;
      const int col = idx < L.KL ? ug_in_col(idx, lane >> 5, C, n_emb, L.KL) : -1;
      if (col >= 0) w = w0[(32 * o + (lane & 31)) * mlp_in + col];
    } else {                                                   // layer 2: k-step st = (o', q)
      const int st = step - L.KB1;
      w = w1[(32 * o + (lane & 31)) * 128 + ug_feat_of(st >> 1, 8 * (st & 1) + e, lane >> 5)];
    }
    const __bf16 hh = (__bf16)w;
    const float r1 = w - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const __bf16 ll = (__bf16)(r1 - (float)mm);
    const __bf16 pick = part == 0 ? hh : (part == 1 ? mm : ll);
    bf[i] = __builtin_bit_cast(unsigned short, pick);
  }
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
    if (i >= L.offB1) {
      out[L.bfB1 + (i - L.offB1)] = v;  // tail copy for the bf16 image
      // fp16x2 tail: biases carry the accumulator scale of their layer, W3 undoes layer 2's
      const float f = i < L.offB2 ? sc.sW1 * sc.sX1 : (i < L.offW3 ? sc.sW2 * sc.sX2 : (i < L.offb3 ? 1.f / (sc.sW2 * sc.sX2) : 1.f));
      if (i >= L.offW3 && i < L.offb3) {
        // fp16x2 image: W3 DENSE, [h][64 rows][3] -- four rows are three ds_read_b128 (12 LDS cycles) instead of four
        // ds_read_b96 (32: the 12-byte read is serviced 8 lanes at a time, MI355X_MICROARCH.md section LDS); the last quarter
        // of the 512-float region stays unused
        const int q = i - L.offW3, c = q & 3, row = q >> 2;       // row = h * 64 + (o * 16 + r)
        if (c < 3) out[L.hxW3 + row * 3 + c] = v * f;
      } else {
        out[L.hxB1 + (i - L.offB1)] = v * f;
      }
    }
  }
}

#ifdef UG_SHADE_PROF
__device__ unsigned long long g_shade_prof[16];   // [8..15]: rgbnet phases of the producer / consumer kernel
extern "C" int ugx_shade_prof_read(unsigned long long *host8) {
  UG_HIP(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_shade_prof), 64));
  unsigned long long z[8] = {0};
  UG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_shade_prof), z, 64));
  return 0;
}
extern "C" int ugx_shade_prof2_read(unsigned long long *host8) {
  UG_HIP(hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_shade_prof), 64, 64));
  unsigned long long z[8] = {0};
  UG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_shade_prof), z, 64, 64));
  return 0;
}
#define UG_PROF_INIT(pr) ug_prof pr; pr.t = __builtin_amdgcn_s_memtime(); for (int i_ = 0; i_ < 8; ++i_) pr.acc[i_] = 0;
#define UG_PROF_FLUSH(pr) if (ug_lane() == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_shade_prof[i_], pr.acc[i_]);
#else
#define UG_PROF_INIT(pr) ug_prof pr;
#define UG_PROF_FLUSH(pr)
#endif

// persistent shade kernel over a work list written by k_march (two-kernel path)
template <int F, int C, int PE, int NW, int BF>
__global__ void __launch_bounds__(NW * 64, NW / 4)
k_shade_mlp(ug_shade_args a, const float *__restrict__ viewdirs, const float *__restrict__ k0b,
            const float *__restrict__ mlp, ug_ws_view ws, float *__restrict__ rgb_marched,
            int32_t *__restrict__ tile_counter) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const ug_mlp_lds M = ug_mlp_stage<C, PE, BF>(lds, mlp, a.residual);
  float *scr = lds + ug_mlp_lds_floats<C, PE, BF>() + (threadIdx.x >> 6) * ug_wave_scratch_floats<C, PE, BF>();
  int victim = 0;
  UG_PROF_INIT(prof)
  for (;;) {
    const int64_t tile = ug_next_tile(tile_counter, ws.n_tiles, blockIdx.x & 7, victim);
    if (tile < 0) break;
    ug_shade_tile<F, C, PE, BF>(a, viewdirs, k0b, M, tile, ws.count[tile], ws.ent + tile * ws.cap,
                                ws.slot + tile * ws.cap, scr, rgb_marched, prof);
  }
  UG_PROF_FLUSH(prof)
}

// producer / consumer shade kernels (ugrid_shade_pc.h): C = 12 quad bricks, fp16x2 rgbnet; NPAIR gather waves + NPAIR rgbnet
// waves per workgroup, one workgroup per CU.  <4, 4 slots, 6 in flight, 4-tile pass> = 8 waves of 256 VGPRs;
// <6, 2 slots, 3 in flight, lean pass> = 12 waves of <= 168 VGPRs
template <int F, int PE, int NPAIR, int SLOTS, int NBL, int MODE>
__global__ void __launch_bounds__(NPAIR * 128, (NPAIR * 2 + 3) / 4)
k_shade_pc(ug_shade_args a, const float *__restrict__ viewdirs, const float *__restrict__ k0b,
           const float *__restrict__ mlp, ug_ws_view ws, float *__restrict__ rgb_marched,
           int32_t *__restrict__ tile_counter) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int MLPF = ug_mlp_lds_floats<12, PE, 2>();
  float *pairs = lds + MLPF;
  if (threadIdx.x < 4 * NPAIR)      // head / tail counters of the rings
    ((int *)(pairs + (threadIdx.x >> 2) * UG_PC_PAIR_FLOATS(SLOTS) + SLOTS * UG_PC_SLOT_FLOATS))[threadIdx.x & 3] = 0;
  const ug_mlp_lds M = ug_mlp_stage<12, PE, 2>(lds, mlp, a.residual);     // ends with __syncthreads()
  const int wv = threadIdx.x >> 6, pair = wv % NPAIR;
  float *ring = pairs + pair * UG_PC_PAIR_FLOATS(SLOTS);
  const unsigned ctl = ug_lds_off(ring + SLOTS * UG_PC_SLOT_FLOATS);
#ifdef UG_SHADE_PROF
  unsigned long long *pstat = g_shade_prof;
#else
  unsigned long long *pstat = nullptr;
#endif
  if (wv < NPAIR) {
    ug_pc_producer<F, NBL, SLOTS>(a, k0b, ws, rgb_marched, tile_counter, ring, ctl, pstat);
  } else {
    float *scr = pairs + NPAIR * UG_PC_PAIR_FLOATS(SLOTS) + pair * ug_pc_consumer_scratch_floats<PE>();
    ug_pc_consumer<PE, SLOTS, MODE>(a, viewdirs, M, rgb_marched, ring, ctl, scr, pstat, ws.emb);
  }
}

// 4 + 8 geometry (round 4): waves 0-3 = producers (one per SIMD), waves 4-11 = consumers (two per SIMD), producer p feeds
// consumers 2p and 2p + 1 (ug_pc_producer2).  Why: a consumer's rgbnet chain is latency-bound (12 k cycles per pass against 4.2 k
// of matrix-pipe issue, profiles/r03/shade_pc12_phases.txt), the 6 + 6 geometry puts 1, 1, 2, 2 consumers on the four SIMDs, and
// the gather side is bound by the CU's shared vector-memory path, not by the number of waves that issue the loads.
template <int F, int PE, int SLOTS, int NBL, int NP = 4, int NC = 8>
__global__ void __launch_bounds__(768, 1)
k_shade_pc48(ug_shade_args a, const float *__restrict__ viewdirs, const float *__restrict__ k0b,
             const float *__restrict__ mlp, ug_ws_view ws, float *__restrict__ rgb_marched,
             int32_t *__restrict__ tile_counter) {
  static_assert(NP + NC == 12 && NC >= NP && NC <= 2 * NP, "12 waves; every producer feeds one or two consumers");
  constexpr int ND = NC - NP;       // producers 0 .. ND-1 feed two consumers (2p, 2p + 1), the others one (ND + p)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int MLPF = ug_mlp_lds_floats<12, PE, 2>();
  float *rings = lds + MLPF;
  if (threadIdx.x < 4 * NC)         // head / tail counters of the rings
    ((int *)(rings + (threadIdx.x >> 2) * UG_PC_PAIR_FLOATS(SLOTS) + SLOTS * UG_PC_SLOT_FLOATS))[threadIdx.x & 3] = 0;
  const ug_mlp_lds M = ug_mlp_stage<12, PE, 2>(lds, mlp, a.residual);     // ends with __syncthreads()
  const int wv = threadIdx.x >> 6;
  if (wv < ND) {
    float *r0 = rings + (2 * wv) * UG_PC_PAIR_FLOATS(SLOTS), *r1 = r0 + UG_PC_PAIR_FLOATS(SLOTS);
    ug_pc_producer2<F, NBL, SLOTS, true>(a, k0b, ws, rgb_marched, tile_counter, r0, ug_lds_off(r0 + SLOTS * UG_PC_SLOT_FLOATS), r1,
                                         ug_lds_off(r1 + SLOTS * UG_PC_SLOT_FLOATS));
  } else if (wv < NP) {
    float *r0 = rings + (ND + wv) * UG_PC_PAIR_FLOATS(SLOTS);
    ug_pc_producer2<F, NBL + UG_PC57_EXTRA, SLOTS, false>(a, k0b, ws, rgb_marched, tile_counter, r0, ug_lds_off(r0 + SLOTS * UG_PC_SLOT_FLOATS), nullptr, 0u);
  } else {
    const int c = wv - NP;
    float *ring = rings + c * UG_PC_PAIR_FLOATS(SLOTS);
    float *scr = rings + NC * UG_PC_PAIR_FLOATS(SLOTS) + c * UG_ACC_SCRATCH_FLOATS;
    ug_pc_consumer<PE, SLOTS, 2>(a, viewdirs, M, rgb_marched, ring, ug_lds_off(ring + SLOTS * UG_PC_SLOT_FLOATS), scr, nullptr, ws.emb);
  }
}

// the view-direction embedding of every ray, once per frame: row = two halves of 16 floats, 14 used, split like the rgbnet's
// first-layer inputs (ug_pc_consumer: emb[h * EH + e]); rays past the end repeat the last one (the consumers never publish them)
template <int PE>
__global__ void __launch_bounds__(256)
k_view_emb(const float *__restrict__ viewdirs, int64_t n_rays, int64_t n_rows, float *__restrict__ emb_rows) {
  constexpr int NEMB = 3 + 6 * PE, EH = (NEMB + 1) / 2;
  static_assert(EH <= UG_EMB_ROW / 2, "row too small");
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const int64_t ray = row < n_rays ? row : n_rays - 1;
  const float vx = viewdirs[3 * ray], vy = viewdirs[3 * ray + 1], vz = viewdirs[3 * ray + 2];
  float emb[UG_EMB_ROW];
#pragma unroll
  for (int e = 0; e < UG_EMB_ROW; ++e) emb[e] = 0.f;
  float lin[2 * EH];
  lin[0] = vx; lin[1] = vy; lin[2] = vz;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float v = ax == 0 ? vx : (ax == 1 ? vy : vz);
#pragma unroll
    for (int k = 0; k < PE; ++k) {
      float s_, c_;
      ug_sincos(v * (float)(1 << k), &s_, &c_);
      lin[3 + ax * PE + k] = s_;
      lin[3 + 3 * PE + ax * PE + k] = c_;
    }
  }
#pragma unroll
  for (int e = NEMB; e < 2 * EH; ++e) lin[e] = 0.f;
#pragma unroll
  for (int e = 0; e < 2 * EH; ++e) emb[(e / EH) * (UG_EMB_ROW / 2) + (e % EH)] = lin[e];
  float4 *dst = (float4 *)(emb_rows + row * UG_EMB_ROW);
#pragma unroll
  for (int q = 0; q < UG_EMB_ROW / 4; ++q) dst[q] = make_float4(emb[4 * q], emb[4 * q + 1], emb[4 * q + 2], emb[4 * q + 3]);
}

#ifdef UG_EXPERIMENTS
// 16-wave variant on 16x16x32 MFMA tiles (ug_shade_tile16): C = 12, PE = 4, fp16x2 arithmetic
template <int F>
__global__ void __launch_bounds__(1024, 1)
k_shade_mlp16(ug_shade_args a, const float *__restrict__ viewdirs, const float *__restrict__ k0b,
              const float *__restrict__ mlp, ug_ws_view ws, float *__restrict__ rgb_marched,
              int32_t *__restrict__ tile_counter, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const ug_mlp16_lds M = ug_mlp16_stage(lds, mlp);
  float *scr = lds + ug_mlp16_lds_floats() + (threadIdx.x >> 6) * UG_ACC16_SCRATCH_FLOATS;
  int victim = 0;
  UG_PROF_INIT(prof)
  for (;;) {
    const int64_t tile = ug_next_tile(tile_counter, ws.n_tiles, blockIdx.x & 7, victim);
    if (tile < 0) break;
    ug_shade_tile16<F>(a, viewdirs, k0b, M, tile, ws.count[tile], ws.ent + tile * ws.cap, ws.slot + tile * ws.cap, scr,
                       rgb_marched, prof, dbg);
  }
  UG_PROF_FLUSH(prof)
}
#endif

// Single-launch render: every persistent wave marches a 64-ray tile and immediately shades the survivors
// it found (its list lives in that wave's private scratch slot and is still L2-resident).  Waves of one CU
// sit in different phases, so the VALU-bound march of some overlaps the MFMA-bound rgbnet of others.
template <int F, bool L2, int C, int PE, int NW, int BF>
__global__ void __launch_bounds__(NW * 64, NW / 4)
k_render_fused(ug_march_args am, ug_shade_args as, const float *__restrict__ rays_o,
               const float *__restrict__ rays_d, const float *__restrict__ viewdirs,
               const float *__restrict__ t_table, const float *__restrict__ s_table,
               const float *__restrict__ dens_bricks, const float *__restrict__ k0b,
               const float *__restrict__ mlp, float *__restrict__ alphainv_last, float *__restrict__ depth,
               float *__restrict__ rgb_marched, float4 *__restrict__ scratch_ent,
               uint8_t *__restrict__ scratch_slot, unsigned long long *__restrict__ survivors_total,
               int32_t *__restrict__ tile_counter, int64_t n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const ug_mlp_lds M = ug_mlp_stage<C, PE, BF>(lds, mlp);
  float *scr = lds + ug_mlp_lds_floats<C, PE, BF>() + (threadIdx.x >> 6) * ug_wave_scratch_floats<C, PE, BF>();
  const int64_t cap = (int64_t)UG_WAVE * am.S;
  const int64_t wslot = (int64_t)blockIdx.x * NW + (threadIdx.x >> 6);
  float4 *__restrict__ ent = scratch_ent + wslot * cap;
  uint8_t *__restrict__ slot = scratch_slot + wslot * cap;
  int victim = 0;
  long long total = 0;
  for (;;) {
    const int64_t tile = ug_next_tile(tile_counter, n_tiles, blockIdx.x & 7, victim);
    if (tile < 0) break;
    const int count = ug_march_tile<F, L2>(am, rays_o, rays_d, t_table, s_table, dens_bricks, alphainv_last,
                                           depth, tile, ent, slot);
    // The list was written by this wave into a scratch slot it re-uses for every tile: its stores are
    // write-through (they are in L2 once vmcnt drains), but this CU's vector L1 may still hold the slot's lines
    // from the previous tile.  Drain the stores, then invalidate the L1 (agent-scope acquire = buffer_inv sc1).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ug_prof prof_unused;
    ug_shade_tile<F, C, PE, BF>(as, viewdirs, k0b, M, tile, count, ent, slot, scr, rgb_marched, prof_unused);
    total += count;
  }
  if (ug_lane() == 0 && total) atomicAdd(survivors_total, (unsigned long long)total);
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
extern "C" int64_t ugrid_mlp_packed_bytes(int32_t k0_channels, int32_t viewbase_pe) {
  return (int64_t)sizeof(float) * ug_mlp_lay(k0_channels, 3 + 6 * viewbase_pe).total4;
}

// largest power of two <= v (v > 0, finite)
static inline float ug_pow2_floor(double v) { return (float)std::ldexp(1.0, (int)std::floor(std::log2(v))); }

// fp16x2 scales from HOST copies of the first two layers (pure host arithmetic: unit-testable without a GPU).
// Interval bounds are propagated through layer 1 (|k0| <= k0_absmax, |view embedding| <= 1); every scale is the
// largest power of two that keeps the largest scaled operand <= 2^15.  Returns 1 when the mode is usable.
extern "C" int ugrid_mlp_fp16x2_scales(const float *h_w0, const float *h_b0, const float *h_w1, int32_t k0_channels,
                                       int32_t viewbase_pe, float k0_absmax, float *scales4) {
  const int C = k0_channels, n_emb = 3 + 6 * (int)viewbase_pe, mlp_in = C + n_emb;
  double m1 = 0, m2 = 0, B1 = 0;
  bool finite = true;   // NaN / inf weights: comparisons would silently skip them
  const double fb = k0_absmax > 0 ? (double)k0_absmax : 0.0;
  for (int n = 0; n < 128; ++n) {
    double acc = std::fabs((double)h_b0[n]);
    m1 = acc > m1 ? acc : m1;   // the 16x16x32 image stores the layer-1 bias as one more weight column (x = 1)
    for (int k = 0; k < mlp_in; ++k) {
      const double a = std::fabs((double)h_w0[(size_t)n * mlp_in + k]);
      finite = finite && std::isfinite(a);
      m1 = a > m1 ? a : m1;
      acc += a * (k < C ? fb : 1.0);
    }
    finite = finite && std::isfinite(acc);
    B1 = acc > B1 ? acc : B1;
  }
  for (size_t i = 0; i < (size_t)128 * 128; ++i) {
    const double a = std::fabs((double)h_w1[i]);
    finite = finite && std::isfinite(a);
    m2 = a > m2 ? a : m2;
  }
  const double B0 = fb > 1.0 ? fb : 1.0;
  ug_mlp_scales sc = {1.f, 1.f, 1.f, 1.f};
  bool ok = finite && k0_absmax > 0 && std::isfinite(fb) && std::isfinite(B1) && std::isfinite(m1) && std::isfinite(m2) &&
            m1 > 1e-30 && m2 > 1e-30 && B1 > 1e-30;
  if (ok) {
    sc.sX1 = ug_pow2_floor(32768.0 / B0);
    sc.sW1 = ug_pow2_floor(32768.0 / m1);
    sc.sX2 = ug_pow2_floor(32768.0 / B1);
    sc.sW2 = ug_pow2_floor(32768.0 / m2);
    // keep every scale (and the products that scale the biases) comfortably inside fp32's exponent range
    const float lo = 1.0f / 1024.0f, hi = 1099511627776.0f;  // 2^-10 .. 2^40
    ok = sc.sX1 >= lo && sc.sX2 >= lo && sc.sW1 >= lo && sc.sW2 >= lo && sc.sX1 <= hi && sc.sX2 <= hi &&
         sc.sW1 <= hi && sc.sW2 <= hi;
    if (!ok) sc = {1.f, 1.f, 1.f, 1.f};
  }
  scales4[0] = sc.sX1; scales4[1] = sc.sW1; scales4[2] = sc.sX2; scales4[3] = sc.sW2;
  return ok ? 1 : 0;
}

extern "C" int ugrid_pack_mlp(const float *w0, const float *b0, const float *w1, const float *b1,
                              const float *w2, const float *b2, int32_t k0_channels, int32_t viewbase_pe,
                              int32_t width, float k0_absmax, float *packed, int32_t *best_mode,
                              ugrid_stream_t s) {
  if (width != 128) return (int)hipErrorInvalidValue;
  const int C = k0_channels, n_emb = 3 + 6 * (int)viewbase_pe, mlp_in = C + n_emb;
  std::vector<float> h0((size_t)128 * mlp_in), hb0(128), h1((size_t)128 * 128);
  UG_HIP(hipMemcpyAsync(h0.data(), w0, h0.size() * sizeof(float), hipMemcpyDeviceToHost, ST(s)));
  UG_HIP(hipMemcpyAsync(hb0.data(), b0, hb0.size() * sizeof(float), hipMemcpyDeviceToHost, ST(s)));
  UG_HIP(hipMemcpyAsync(h1.data(), w1, h1.size() * sizeof(float), hipMemcpyDeviceToHost, ST(s)));
  UG_HIP(hipStreamSynchronize(ST(s)));
  float sc4[4];
  const int ok = ugrid_mlp_fp16x2_scales(h0.data(), hb0.data(), h1.data(), k0_channels, viewbase_pe, k0_absmax, sc4);
  const ug_mlp_scales sc = {sc4[0], sc4[1], sc4[2], sc4[3]};
  hipLaunchKernelGGL(k_pack_mlp, dim3(64), dim3(256), 0, ST(s), w0, b0, w1, b1, w2, b2, C, n_emb, sc, packed);
  UG_LAUNCH_CHECK();
  // (viewbase_pe > 4: the fp16x2 kernels' LDS geometry does not fit, see ug_shade_launch)
  if (best_mode) *best_mode = (ok && viewbase_pe <= 4) ? UGRID_MLP_FP16X2 : UGRID_MLP_BF16X3;
  return 0;
}

// ---- single-launch fused render -----------------------------------------------------------------
#define UG_FUSED_NW 12  // slots are sized for the largest variant
#define UG_FUSED_MAX_WGS 256

extern "C" int64_t ugrid_render_fused_ws_bytes(int32_t n_samples) {
  const int64_t slots = (int64_t)UG_FUSED_MAX_WGS * UG_FUSED_NW, cap = (int64_t)UG_WAVE * n_samples;
  return 256 + ug_align256(slots * cap * 16) + ug_align256(slots * cap);
}


template <int F, bool L2, int C, int PE, int NW, int BF>
static int ug_fused_launch_nw(const ug_march_args &am, const ug_shade_args &as, const float *rays_o,
                           const float *rays_d, const float *viewdirs, const float *t_table,
                           const float *s_table, const float *dens_bricks, const float *k0b, const float *mlp,
                           float *alphainv_last, float *depth, float *rgb, void *ws_mem, hipStream_t st) {
  const int lds_bytes = ug_shade_lds_bytes<C, PE, BF, NW>();
  UG_SET_DYN_LDS((k_render_fused<F, L2, C, PE, NW, BF>), lds_bytes);
  const int64_t n_tiles = (am.n_rays + UG_WAVE - 1) / UG_WAVE;
  const int64_t slots = (int64_t)UG_FUSED_MAX_WGS * NW, cap = (int64_t)UG_WAVE * am.S;
  char *base = (char *)ws_mem;
  UG_ZERO_WORDS(base, 64, st);  // 8 tile counters @0, survivor total @64
  float4 *ent = (float4 *)(base + 256);
  uint8_t *slot = (uint8_t *)(base + 256 + ug_align256(slots * cap * 16));
  int64_t wgs = (n_tiles + NW - 1) / NW;
  if (wgs > UG_FUSED_MAX_WGS) wgs = UG_FUSED_MAX_WGS;
  wgs = (wgs + 7) / 8 * 8;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_render_fused<F, L2, C, PE, NW, BF>), dim3((unsigned)wgs), dim3(NW * 64), lds_bytes,
                     st, am, as, rays_o, rays_d, viewdirs, t_table, s_table, dens_bricks, k0b, mlp, alphainv_last,
                     depth, rgb, ent, slot, (unsigned long long *)(base + 64), (int32_t *)base, n_tiles);
  UG_LAUNCH_CHECK();
  return 0;
}

template <int F, bool L2, int C, int PE>
static int ug_fused_launch(const ug_march_args &am, const ug_shade_args &as, const float *rays_o,
                           const float *rays_d, const float *viewdirs, const float *t_table,
                           const float *s_table, const float *dens_bricks, const float *k0b, const float *mlp,
                           float *alphainv_last, float *depth, float *rgb, void *ws_mem, int mlp_mode,
                           hipStream_t st) {
  // the single-launch variant is kept for experiments only (bf16x3 or fp32 MFMA; fp16x2 is not instantiated)
  if (mlp_mode != UGRID_MLP_FP32)
    return ug_fused_launch_nw<F, L2, C, PE, 8, 1>(am, as, rays_o, rays_d, viewdirs, t_table, s_table, dens_bricks,
                                                     k0b, mlp, alphainv_last, depth, rgb, ws_mem, st);
  return ug_fused_launch_nw<F, L2, C, PE, 8, 0>(am, as, rays_o, rays_d, viewdirs, t_table, s_table, dens_bricks,
                                                    k0b, mlp, alphainv_last, depth, rgb, ws_mem, st);
}

extern "C" int ugrid_render_fused(const ugrid_render_params *p, const float *rays_o, const float *rays_d,
                                  const float *viewdirs, const float *t_table, const float *s_table,
                                  const float *density_bricks, const float *k0_bricks, const float *mlp_packed,
                                  float *alphainv_last, float *depth, float *rgb_marched, void *ws_mem,
                                  ugrid_stream_t s) {
  if (p->n_rays <= 0) return 0;
  if (p->mlp_in == 0 || p->mlp_width != 128 || p->mlp_in != p->k0_channels + 3 + 6 * p->viewbase_pe)
    return (int)hipErrorNotSupported;  // rgbnet-less models use the two-kernel path
  if (p->mlp_mode & UGRID_MLP_RESIDUAL) return (int)hipErrorNotSupported;   // (the residual epilogue exists in the two-kernel path only)
  ug_march_args am;
  const int rc = ug_fill_march_args(p, am);
  if (rc) return rc;
  ug_shade_args as;
  ug_fill_shade_args(p, as);
#define UG_FUSED_CASE(F_, C_, PE_)                                                                        \
  if (p->freq_num == F_ && p->k0_channels == C_ && p->viewbase_pe == PE_) {                               \
    if (p->norm_l2)                                                                                       \
      return ug_fused_launch<F_, true, C_, PE_>(am, as, rays_o, rays_d, viewdirs, t_table, s_table,       \
                                                density_bricks, k0_bricks, mlp_packed, alphainv_last,     \
                                                depth, rgb_marched, ws_mem, p->mlp_mode, ST(s));          \
    return ug_fused_launch<F_, false, C_, PE_>(am, as, rays_o, rays_d, viewdirs, t_table, s_table,        \
                                               density_bricks, k0_bricks, mlp_packed, alphainv_last,      \
                                               depth, rgb_marched, ws_mem, p->mlp_mode, ST(s));           \
  }
  UG_FUSED_CASE(3, 12, 4)
  UG_FUSED_CASE(4, 12, 4)
#undef UG_FUSED_CASE
  return (int)hipErrorNotSupported;
}

extern "C" int ugrid_render_fused_stats(const void *ws_mem, int64_t *d_stats, ugrid_stream_t s) {
  return (int)hipMemcpyAsync(d_stats, (const char *)ws_mem + 64, sizeof(int64_t), hipMemcpyDeviceToDevice, ST(s));
}


extern "C" int ugrid_tune(const char *key, int value) {
  if (!key) return (int)hipErrorInvalidValue;
  if (!strcmp(key, "march_waves")) return ug_set_march_waves(value) ? (int)hipErrorInvalidValue : 0;
  if (!strcmp(key, "tv_xcd")) return ug_set_tv_xcd(value) ? (int)hipErrorInvalidValue : 0;
  if (!strcmp(key, "shade_pc") && value >= 0 && value <= 5) { g_shade_pc = value; return 0; }
#ifdef UG_EXPERIMENTS
  if (!strcmp(key, "shade16") && (value == 0 || value == 1)) { g_shade16 = value; return 0; }
  if (!strcmp(key, "shade_dbg") && value >= 0 && value < 4) { g_shade_dbg = value; return 0; }
#endif
  return (int)hipErrorInvalidValue;
}

template <int F, int C, int PE, int NW, int BF>
static int ug_shade_launch_nw(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                              ug_ws_view ws, float *rgb, int32_t *counter, hipStream_t st) {
  const int lds_bytes = ug_shade_lds_bytes<C, PE, BF, NW>();
  UG_SET_DYN_LDS((k_shade_mlp<F, C, PE, NW, BF>), lds_bytes);
  UG_ZERO_WORDS(counter, 8, st);
  // persistent: one workgroup per CU (LDS holds the packed rgbnet image of the mode: 89 / 138 / 93 KB)
  int64_t wgs = (ws.n_tiles + NW - 1) / NW;
  if (wgs > 256) wgs = 256;
  wgs = (wgs + 7) / 8 * 8;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shade_mlp<F, C, PE, NW, BF>), dim3((unsigned)wgs), dim3(NW * 64), lds_bytes, st, a,
                     viewdirs, k0b, mlp, ws, rgb, counter);
  UG_LAUNCH_CHECK();
  return 0;
}


template <int F, int PE, int NPAIR, int SLOTS, int NBL, int MODE>
static int ug_shade_pc_launch(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                              ug_ws_view ws, float *rgb, int32_t *counter, hipStream_t st) {
  const int lds_bytes = ug_pc_lds_bytes<PE, NPAIR, SLOTS>();
  if (lds_bytes > 160 * 1024) return (int)hipErrorInvalidValue;   // the CU's LDS
  UG_SET_DYN_LDS((k_shade_pc<F, PE, NPAIR, SLOTS, NBL, MODE>), lds_bytes);
  UG_ZERO_WORDS(counter, 8, st);
  if constexpr (MODE == 2) {      // A/B arm: the 6 + 6 geometry with the consumers of the 4 + 8 / 5 + 7 ones (embedding rows from global memory)
    const int64_t n_rows = ws.n_tiles * UG_WAVE;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_view_emb<PE>), dim3(ug_blocks(n_rows, 256)), dim3(256), 0, st, viewdirs, a.n_rays, n_rows, ws.emb);
  }
  // persistent, one workgroup per CU: NPAIR producer waves pull tiles, so a workgroup covers >= NPAIR tiles
  int64_t wgs = (ws.n_tiles + NPAIR - 1) / NPAIR;
  if (wgs > 256) wgs = 256;
  wgs = (wgs + 7) / 8 * 8;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shade_pc<F, PE, NPAIR, SLOTS, NBL, MODE>), dim3((unsigned)wgs), dim3(NPAIR * 128), lds_bytes, st,
                     a, viewdirs, k0b, mlp, ws, rgb, counter);
  UG_LAUNCH_CHECK();
  return 0;
}

template <int F, int PE, int SLOTS, int NBL, int NP, int NC>
static int ug_shade_pc48_launch(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                                ug_ws_view ws, float *rgb, int32_t *counter, hipStream_t st) {
  const int lds_bytes = ug_pc48_lds_bytes<PE, SLOTS, NC>();
  if (lds_bytes > 160 * 1024) return (int)hipErrorInvalidValue;
  UG_SET_DYN_LDS((k_shade_pc48<F, PE, SLOTS, NBL, NP, NC>), lds_bytes);
  UG_ZERO_WORDS(counter, 8, st);
  const int64_t n_rows = ws.n_tiles * UG_WAVE;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_view_emb<PE>), dim3(ug_blocks(n_rows, 256)), dim3(256), 0, st, viewdirs, a.n_rays, n_rows, ws.emb);
  int64_t wgs = (ws.n_tiles + NC - 1) / NC;    // NC tile streams per workgroup
  if (wgs > 256) wgs = 256;
  wgs = (wgs + 7) / 8 * 8;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shade_pc48<F, PE, SLOTS, NBL, NP, NC>), dim3((unsigned)wgs), dim3(768), lds_bytes, st,
                     a, viewdirs, k0b, mlp, ws, rgb, counter);
  UG_LAUNCH_CHECK();
  return 0;
}

#ifdef UG_EXPERIMENTS
template <int F>
static int ug_shade16_launch(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                             ug_ws_view ws, float *rgb, int32_t *counter, hipStream_t st) {
  const int lds_bytes = (int)sizeof(float) * (ug_mlp16_lds_floats() + 16 * UG_ACC16_SCRATCH_FLOATS);
  UG_SET_DYN_LDS((k_shade_mlp16<F>), lds_bytes);
  UG_ZERO_WORDS(counter, 8, st);
  int64_t wgs = (ws.n_tiles + 15) / 16;
  if (wgs > 256) wgs = 256;
  wgs = (wgs + 7) / 8 * 8;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shade_mlp16<F>), dim3((unsigned)wgs), dim3(1024), lds_bytes, st, a, viewdirs, k0b,
                     mlp, ws, rgb, counter, g_shade_dbg);
  UG_LAUNCH_CHECK();
  return 0;
}

#endif

template <int F, int C, int PE>
static int ug_shade_launch(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                           ug_ws_view ws, float *rgb, int32_t *counter, int mlp_mode, hipStream_t st) {
#ifdef UG_EXPERIMENTS
  if constexpr (C == 12 && PE == 4) {
    if (mlp_mode == UGRID_MLP_FP16X2 && g_shade16) return ug_shade16_launch<F>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
  }
#endif
  // fp16x2 keeps a per-wave view-embedding table in LDS next to the rgbnet image: with viewbase_pe = 8 that is 14 KB per
  // consumer wave on top of a 99 KB image -- no 8-wave geometry fits the CU's 160 KB.  ugrid_pack_mlp reports bf16x3 as the
  // best mode for such networks and the fp16x2 kernels are not instantiated for them.
  if constexpr (PE > 4) {
    if (mlp_mode == UGRID_MLP_FP16X2) return (int)hipErrorInvalidValue;
  }
  if constexpr (C == 12 && PE <= 4) {
    if constexpr (F <= 3 && PE == 4) {
      if (mlp_mode == UGRID_MLP_FP16X2 && g_shade_pc == 3)
        return ug_shade_pc48_launch<F, PE, UG_PC48_SLOTS, UG_PC48_NBL, 4, 8>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
      if (mlp_mode == UGRID_MLP_FP16X2 && g_shade_pc == 5)
        return ug_shade_pc_launch<F, PE, 6, 2, UG_PC12_NBL, 2>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
      if (mlp_mode == UGRID_MLP_FP16X2 && g_shade_pc == 4)
        return ug_shade_pc48_launch<F, PE, UG_PC48_SLOTS, UG_PC48_NBL, 5, 7>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
    }
    if constexpr (F <= 3) {      // the producers' set-up state grows with the level count: F >= 4 does not fit 168 VGPRs
      if (mlp_mode == UGRID_MLP_FP16X2 && g_shade_pc >= 2)
        return ug_shade_pc_launch<F, PE, 6, 2, UG_PC12_NBL, 1>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
    }
    if (mlp_mode == UGRID_MLP_FP16X2 && g_shade_pc >= 1)
      return ug_shade_pc_launch<F, PE, 4, 4, UG_PC_NBL(F), 0>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
  }
  // every variant runs 8 waves per workgroup (2 per SIMD, <= 256 registers each).  A 12-wave bf16x3 build needed
  // spills and gained 4 %; it is not instantiated.
  if constexpr (PE <= 4) {
    if (mlp_mode == UGRID_MLP_FP16X2) return ug_shade_launch_nw<F, C, PE, 8, 2>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
  }
  if (mlp_mode == UGRID_MLP_BF16X3) return ug_shade_launch_nw<F, C, PE, 8, 1>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
  if constexpr (C == 9) return (int)hipErrorNotSupported;      // (the exact-fp32 MFMA variant of this shape needs scratch: not built)
  else return ug_shade_launch_nw<F, C, PE, 8, 0>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
}

extern "C" int ugrid_render_shade(const ugrid_render_params *p, const float *viewdirs, const float *k0_bricks,
                                  const float *mlp_packed, void *ws_mem, float *rgb_marched, ugrid_stream_t s) {
  if (p->n_rays <= 0) return 0;
  ug_ws_view ws = ug_ws_make(ws_mem, p->n_rays, p->n_samples);
  ug_shade_args a;
  ug_fill_shade_args(p, a);
  if (p->mlp_in == 0) {
    if (p->k0_channels != 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_shade_direct, dim3(ug_blocks(ws.n_tiles * UG_WAVE, 256)), dim3(256), 0, ST(s), a,
                       k0_bricks, ws, rgb_marched);
    UG_LAUNCH_CHECK();
    return 0;
  }
  if (p->mlp_width != 128 || p->mlp_in != p->k0_channels + 3 + 6 * p->viewbase_pe) return (int)hipErrorInvalidValue;
  const int mlp_mode = p->mlp_mode & ~UGRID_MLP_RESIDUAL;
  if (mlp_mode < UGRID_MLP_FP32 || mlp_mode > UGRID_MLP_FP16X2) return (int)hipErrorInvalidValue;
  // residual colour: lane (h = 0) must hold channels 0..2 of its survivor, i.e. at least 3 channels per half-brick (C >= 9)
  if ((p->mlp_mode & UGRID_MLP_RESIDUAL) && UG_CH(p->k0_channels) < 3) return (int)hipErrorInvalidValue;
  int32_t *counter = (int32_t *)ws_mem;  // first 256 B of the work list
// the instantiated (F, C, PE) triples: Mip-NeRF-360 *_single.py (configs/default.py:104-124); tankstemple_unbounded/
// truck_single.py:105; FourierGridModel's constructor default fourier_freq_num = 5 (FourierGrid_model.py:137); waymo-style
// rgbnet_dim = 3, viewbase_pe = 2 (configs/waymo/waymo_no_block.py:144-149)
// F = 0: single-level k0 (DirectContractedVoxGO / DenseGrid models, configs/nerf_unbounded/*.py: rgbnet_dim 12)
// viewbase_pe = 8 (configs/waymo/waymo_base.py, configs/mega/*.py; rgbnet_dim 3 in mega/building_no_block.py); rgbnet_dim = 15
// (configs/tankstemple_unbounded/train_single.py); rgbnet_dim = 9 (configs/free_dataset/*.py, whose rgbnet_width = 64 the host
// pads to 128: fourier_render.pad_rgbnet_to_128)
#define UG_SHADE_TRIPLES(X) X(3, 12, 4) X(4, 12, 4) X(5, 12, 4) X(2, 12, 4) X(1, 12, 4) X(0, 12, 4) X(2, 3, 2) X(3, 3, 2) X(3, 12, 8) X(3, 3, 8) X(3, 15, 4) X(3, 9, 4) X(0, 9, 4)
#define UG_SHADE_CASE(F_, C_, PE_)                                                          \
  if (p->freq_num == F_ && p->k0_channels == C_ && p->viewbase_pe == PE_)                   \
    return ug_shade_launch<F_, C_, PE_>(a, viewdirs, k0_bricks, mlp_packed, ws, rgb_marched, counter, mlp_mode, ST(s));
  UG_SHADE_TRIPLES(UG_SHADE_CASE)
#undef UG_SHADE_CASE
  return (int)hipErrorNotSupported;
}

// 1 when ugrid_render_shade has an rgbnet instantiation (depth 3, width 128) for this (fourier_freq_num, k0 channels,
// viewbase_pe) triple -- callers pick the composed path otherwise (fourier_render.ComposedFourierGridRenderer)
extern "C" int ugrid_shade_supported(int32_t freq_num, int32_t k0_channels, int32_t viewbase_pe) {
#define UG_SHADE_HAS(F_, C_, PE_) if (freq_num == F_ && k0_channels == C_ && viewbase_pe == PE_) return 1;
  UG_SHADE_TRIPLES(UG_SHADE_HAS)
#undef UG_SHADE_HAS
  return 0;
}

extern "C" int ugrid_render_stats(void *ws_mem, int64_t n_rays, int32_t S, int64_t *d_stats, ugrid_stream_t s) {
  ug_ws_view ws = ug_ws_make(ws_mem, n_rays, S);
  UG_ZERO_WORDS(d_stats, 2, ST(s));
  hipLaunchKernelGGL(k_ws_stats, dim3(64), dim3(256), 0, ST(s), ws.count, ws.n_tiles, d_stats);
  UG_LAUNCH_CHECK();
  return 0;
}
