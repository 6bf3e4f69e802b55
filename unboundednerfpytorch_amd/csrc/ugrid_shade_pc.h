// ugrid_shade_pc.h -- producer / consumer form of the shade kernel (C = 12 quad bricks, fp16x2 rgbnet), gfx950.
//
// Why: in k_shade_mlp every wave runs gather -> rgbnet -> accumulation strictly in
// series (22 k ticks per 32-survivor pass at 2 waves per SIMD, every unit under 40 % busy); overlapping the gather of pass
// i+1 with the rgbnet of pass i INSIDE a wave needs ~100 more VGPRs than the 256 a wave has beside the 128 accumulator
// registers (cross-pass prefetch: 47-120 spilled registers, 10.2 ms).  Here the two halves run in DIFFERENT waves:
//
//   waves 0..3  PRODUCERS: pull 64-ray tiles (XCD-affine atomic counters), walk the tile's survivor list 32 at a time and
//               gather the k0 features (ug_k0_gather_quad: lane quads, volatile-asm load pipeline, cell polynomials) --
//               with no accumulators to hold, NBL = 6 items (36 dwordx4) stay in flight per wave; the 32 x 12 features of a
//               pass + the survivors' weights and ray slots go into a slot of the pair's LDS ring;
//   waves 4..7  CONSUMERS: take a slot, copy its 6 + 2 values per lane into registers, hand the slot back at once, and run
//               the fp16x2 MFMA chain + layer 3 + the ordered per-ray accumulation (ug_rgbnet_pass, unchanged arithmetic).
//
// Wave w and w + 4 of a workgroup land on the same SIMD: every SIMD has exactly one wave on its matrix pipe and one wave
// feeding the vector-memory path, so neither competes with a twin for its unit.  The ring (SLOTS slots per pair)
// decouples memory latency from the MFMA chain.  Hand-off = two monotone LDS counters per pair (head: written by the
// producer only, tail: by the consumer only), polled with ds_read + s_sleep; LDS operations of a wave complete in order, a
// counter is published after `s_waitcnt lgkmcnt(0)`.  Results are bit-identical to k_shade_mlp: same gather, same rgbnet,
// same accumulation order (list order inside a tile; a tile is owned by one pair).
#pragma once
#include "ugrid_render.h"

// Two geometries are built (ugrid_shade.hip):
//   k_shade_pc   <F, PE>  8 waves / CU (256 VGPRs each):  4 producers (6 gather items in flight) + 4 consumers running the
//                         hand-scheduled 4-tile rgbnet pass, 4-slot rings;
//   k_shade_pc12 <F, PE> 12 waves / CU (<= 168 VGPRs each): 6 producers (3 items in flight) + 6 consumers running the LEAN
//                         pass (layer 2 two output tiles at a time, ug_rgbnet_pass_lean), 2-slot rings.  Why: the rgbnet chain
//                         of ONE wave is latency-bound (9-11 k ticks per pass against 4.2 k of MFMA issue, phase profiles in
//                         profiles/r03/); three waves per SIMD overlap each other's stalls, two cannot.
// 8-wave geometry: gather items (x 6 dwordx4) in flight per producer wave: 6 at F <= 3; the set-up state grows with the level count
// (4 registers per (round, level)), so F = 4 keeps 5 and F = 5 keeps 4 in flight to stay inside 256 VGPRs without scratch
#define UG_PC_NBL(F) ((F) <= 3 ? 6 : ((F) == 4 ? 5 : 4))
// ring slot (floats): feat [32][12] | w [32] | sl [32] (int) | hdr {tile, count, base, -} (int)
#define UG_PC_FEAT 0
#define UG_PC_W 384
#define UG_PC_SL 416
#define UG_PC_HDR 448
#define UG_PC_SLOT_FLOATS 452
// per pair: ring | ctl {head, tail, -, -}
#define UG_PC_PAIR_FLOATS(SLOTS) ((SLOTS) * UG_PC_SLOT_FLOATS + 4)

template <int PE>
__host__ __device__ static inline int ug_pc_consumer_scratch_floats() {
  return ug_wave_scratch_floats<12, PE, 2>();      // amask | aval | per-tile embedding table
}
template <int PE, int NPAIR, int SLOTS>
__host__ __device__ static inline int ug_pc_lds_bytes() {
  return (int)sizeof(float) * (ug_mlp_lds_floats<12, PE, 2>() + NPAIR * UG_PC_PAIR_FLOATS(SLOTS) + NPAIR * ug_pc_consumer_scratch_floats<PE>());
}

// ---- LDS counters: explicit ds_ instructions on the 32-bit LDS offset (no flat_ access may sneak in: flat operations
// count on vmcnt AND lgkmcnt and would corrupt the producers' hand-counted vmcnt waits)
typedef __attribute__((address_space(3))) const void *ug_lds_cptr;
__device__ __forceinline__ unsigned ug_lds_off(const void *p) { return (unsigned)(uintptr_t)(ug_lds_cptr)p; }
__device__ __forceinline__ int ug_lds_peek(unsigned off) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
// publish: everything this wave wrote to / read from LDS before is complete, then the counter
__device__ __forceinline__ void ug_lds_publish(unsigned off, int v) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" :: "v"(off), "v"(v) : "memory");
}


// Issue priority of the consumer waves (round 4, profiles/r04/shade_priority_ab.txt): the SIMD's arbiter serves the wave with the
// highest s_setprio first.  A consumer raises its priority to UG_PC_PRIO_PHASED (default 3, the maximum) while it feeds the
// matrix pipe -- layers 1 and 2: an MFMA that waits behind the gather waves' address arithmetic leaves the pipe idle -- and drops
// back to 0 for the VALU-only layer 3 / accumulation / slot wait, where the producers' work should win.  Measured on S1: 4.29-4.34
// -> 4.05-4.11 ms (static priorities 1, 2, 3 for the whole consumer: 4.11-4.14).  Scheduling only: results are bit-identical.
// polling intervals of the ring hand-off (s_sleep units of 64 clocks): a waiting producer polls the consumer's counter, a waiting
// consumer the producer's; every poll is ~10 instructions on a SIMD it shares with working waves (A/B: profiles/r04/shade_poll_ab.txt)
#define UG_PC_PRODUCER_SLEEP 2
#define UG_PC_CONSUMER_SLEEP 2
// TIMING-ONLY ablation arms (tools/experiments/ARMS.md "round 6"; never defined in the shipped build -- results are WRONG with any of
// them): each removes one piece of a pass so that its cost on the whole kernel can be measured instead of priced.
//   UG_ABL_NO_EMB_KSTEPS  layer 1 runs its first k-step only (12 of its 36 MFMAs) and the per-pass embedding-table read is skipped:
//                         an UPPER bound of what hoisting the per-ray half of layer 1 can return (the hoist's own table reads excluded)
//   UG_ABL_NO_L3          no layer-3 VALU work (64 relu + 192 FMA + 66 W3 reads): upper bound of "layer 3 off the VALU"
//   UG_ABL_NO_CONSUME     the consumer hands every slot straight back: the producers' (gather) side alone
//   UG_ABL_NO_GATHER      the producer publishes passes without loading a brick: the consumers' (rgbnet) side alone
#define UG_PRIO_HI() __builtin_amdgcn_s_setprio(3)
#define UG_PRIO_LO() __builtin_amdgcn_s_setprio(0)

// ---------------------------------------------------------------------------------------------------------------------
// LEAN fp16x2 rgbnet pass for the 12-wave geometry (<= 168 VGPRs): same products, same accumulation order per output
// element and the same layer-3 / per-ray summation order as ug_rgbnet_pass_h2 -- bit-identical results -- with a smaller
// register footprint:
//   * layer 1 as before (4 output tiles = 64 accumulator registers, the 20 inputs die as they are split);
//   * the hidden activations are rectified and split ONCE into their fp16 (h, l) operand form, in place of the layer-1
//     accumulators (8 k-steps x 8 registers = the same 64 registers);
//   * layer 2 runs TWO of its four output tiles at a time (32 accumulator registers instead of 64), each half followed by
//     its 32 rows of layer 3 on the VALU; the second half re-uses the split operands.
// The wave leaves stalls exposed (dependent MFMAs two apart, layer 3 not overlapped with MFMAs): its two SIMD neighbours
// fill them.
// ---------------------------------------------------------------------------------------------------------------------
struct ug_hpair { f16x8 w[2]; };
__device__ __forceinline__ ug_hpair ug_load_hpair(const f16x8 *__restrict__ Ap, int o0, int part) {
  ug_hpair p;
  p.w[0] = Ap[((o0 + 0) * 2 + part) * 64];
  p.w[1] = Ap[((o0 + 1) * 2 + part) * 64];
  return p;
}

template <int C, int PE>
__device__ __forceinline__ void ug_rgbnet_pass_lean(const float (&x)[(2 * UG_CH(C) + 3 + 6 * PE + 1) / 2], float ww, int sl, bool ok,
                                                    const ug_mlp_lds &M, unsigned *amask, float4 *aval, float &accr, float &accg,
                                                    float &accb) {
  constexpr int CH = UG_CH(C);
  constexpr int NEMB = 3 + 6 * PE;
  constexpr int KL = (2 * CH + NEMB + 1) / 2;
  constexpr int KB1 = (KL + 7) / 8;
  const int lane = ug_lane();
  const int h = lane >> 5, sv = lane & 31;
  UG_PRIO_HI();
  int bo = h * 64;
  asm volatile("" : "+v"(bo));
  const f16x8 *A1h = (const f16x8 *)M.A1, *A2h = (const f16x8 *)M.A2;
  ug_split2 hs[8];                      // layer-2 B operands: relu(layer 1) * c12, split, k-step major
  float hraw[64];                       // layer-1 outputs until they are split (dies 8 values per k-step of the first half)
  {
    // ---- layer 1
    f32x16 acc1[4];
    {
      const float4 *b1p = (const float4 *)(M.B1 + bo);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b = b1p[o * 4 + q];
          acc1[o][4 * q] = b.x; acc1[o][4 * q + 1] = b.y; acc1[o][4 * q + 2] = b.z; acc1[o][4 * q + 3] = b.w;
        }
    }
    ug_hpart wl = ug_load_hpart(A1h + lane, 1);
    ug_split2 xs, xn;
    {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (e < KL) ? x[e < KL ? e : 0] : 0.f;
      xs = ug_split8h(v, M.sx1);
    }
    ug_fence_operands();
#ifdef UG_ABL_NO_EMB_KSTEPS
    constexpr int KB1_RUN = 1;
#else
    constexpr int KB1_RUN = KB1;
#endif
#pragma unroll
    for (int s = 0; s < KB1_RUN; ++s) {
      float vn[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vn[e] = (8 * (s + 1) + e < KL) ? x[(8 * (s + 1) + e < KL) ? 8 * (s + 1) + e : 0] : 0.f;
      ug_mfma3x4(A1h + (s * 8) * 64 + lane, (s + 1 < KB1_RUN ? A1h + ((s + 1) * 8) * 64 : A2h) + lane, xs, vn, M.sx1, xn, acc1, wl);
      xs = xn;
    }
    ug_fence_results();
    // ---- hidden activations -> operand form: k-step 0 now, k-step st + 1 behind the MFMAs of k-step st of the first half
    {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ug_relu(acc1[0][e]);
      hs[0] = ug_split8h(v, M.c12);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) hraw[o * 16 + r] = acc1[o][r];
  }
  ug_fence_operands();
  float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
  for (int pr2 = 0; pr2 < 2; ++pr2) {
    // ---- layer 2, output tiles 2 pr2 and 2 pr2 + 1
    if (pr2 == 1) { UG_PRIO_HI(); }
    f32x16 acc2[2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = ((const float4 *)(M.B2 + bo))[(2 * pr2 + o) * 4 + q];
        acc2[o][4 * q] = b.x; acc2[o][4 * q + 1] = b.y; acc2[o][4 * q + 2] = b.z; acc2[o][4 * q + 3] = b.w;
      }
    ug_hpair wl2 = ug_load_hpair(A2h + lane, 2 * pr2, 1);
    ug_hpair wh2 = ug_load_hpair(A2h + lane, 2 * pr2, 0);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const f16x8 *An = A2h + ((st + 1 < 8 ? st + 1 : st) * 8) * 64 + lane;
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 hh, ll;
      const bool split_next = (pr2 == 0 && st + 1 < 8);
      const float *hv = hraw + 8 * (st + 1 < 8 ? st + 1 : st);
      // two MFMAs per group (tiles 0 / 1 of this half); the next k-step's weights are requested one whole k-step ahead, and
      // in the first half the next k-step's activations are rectified and split behind the MFMAs (4 VALU each)
      UG_MFMA_F16(acc2[0], wl2.w[0], hs[st].h);
      if (split_next) { unsigned h_, l_; ug_split_pair(ug_relu(hv[0]), ug_relu(hv[1]), M.c12, h_, l_); hh[0] = h_; ll[0] = l_; }
      __builtin_amdgcn_sched_barrier(0);
      UG_MFMA_F16(acc2[1], wl2.w[1], hs[st].h);
      if (split_next) { unsigned h_, l_; ug_split_pair(ug_relu(hv[2]), ug_relu(hv[3]), M.c12, h_, l_); hh[1] = h_; ll[1] = l_; }
      const ug_hpair nwl = ug_load_hpair(An, 2 * pr2, 1);
      __builtin_amdgcn_sched_barrier(0);
      UG_MFMA_F16(acc2[0], wh2.w[0], hs[st].l);
      if (split_next) { unsigned h_, l_; ug_split_pair(ug_relu(hv[4]), ug_relu(hv[5]), M.c12, h_, l_); hh[2] = h_; ll[2] = l_; }
      __builtin_amdgcn_sched_barrier(0);
      UG_MFMA_F16(acc2[1], wh2.w[1], hs[st].l);
      if (split_next) { unsigned h_, l_; ug_split_pair(ug_relu(hv[6]), ug_relu(hv[7]), M.c12, h_, l_); hh[3] = h_; ll[3] = l_; }
      __builtin_amdgcn_sched_barrier(0);
      UG_MFMA_F16(acc2[0], wh2.w[0], hs[st].h);
      UG_MFMA_F16(acc2[1], wh2.w[1], hs[st].h);
      const ug_hpair nwh = ug_load_hpair(An, 2 * pr2, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (split_next) { hs[st + 1].h = __builtin_bit_cast(f16x8, hh); hs[st + 1].l = __builtin_bit_cast(f16x8, ll); }
      wl2 = nwl; wh2 = nwh;
    }
    // first rows of this half's layer-3 weights, requested before the MFMAs have drained
    constexpr int W3B = 4;      // rows per batch = three 16-byte reads of the dense W3 image, two batches in registers
    ug_w3x4 w3[2];
    w3[0] = ug_w3_load4(M, bo, 32 * pr2);
    ug_fence_results();
    UG_PRIO_LO();
    // ---- layer 3, rows 32 pr2 .. 32 pr2 + 31 (same order as the 4-tile pass: rows ascending)
#ifdef UG_ABL_NO_L3
    l0 += acc2[0][0]; l1 += acc2[0][1]; l2 += acc2[1][0];
    constexpr int L3_FIRST = 32;
#else
    constexpr int L3_FIRST = 0;
#endif
#pragma unroll
    for (int sb = L3_FIRST; sb < 32; sb += W3B) {
      const int cur = (sb / W3B) & 1;
      if (sb + W3B < 32) w3[cur ^ 1] = ug_w3_load4(M, bo, 32 * pr2 + sb + W3B);
      __builtin_amdgcn_sched_barrier(0);
      {
        const float h0 = ug_relu(acc2[sb >> 4][sb & 15]), h1 = ug_relu(acc2[sb >> 4][(sb & 15) + 1]);
        const float h2 = ug_relu(acc2[sb >> 4][(sb & 15) + 2]), h3 = ug_relu(acc2[sb >> 4][(sb & 15) + 3]);
        UG_W3_FMA4(w3[cur], h0, h1, h2, h3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  l0 = (l0 + __shfl_xor(l0, 32)) + M.b3[0];
  l1 = (l1 + __shfl_xor(l1, 32)) + M.b3[1];
  l2 = (l2 + __shfl_xor(l2, 32)) + M.b3[2];
  UG_RESIDUAL_ADD(M, x, l0, l1, l2)
  const float pr = ww * ug_sigmoid(l0), pg = ww * ug_sigmoid(l1), pb = ww * ug_sigmoid(l2);
  {
    // ordered per-ray sum through LDS (masks pre-cleared, see ug_rgbnet_pass_h2)
    if (ok && h == 0) {
      aval[sv] = make_float4(pr, pg, pb, 0.f);
      atomicOr(&amask[sl], 1u << sv);
    }
    ug_wave_lds_sync();
    unsigned m = amask[lane];
    amask[lane] = 0u;
    while (m) {
      const int kk = __builtin_ctz(m);
      const float4 t = aval[kk];
      accr += t.x; accg += t.y; accb += t.z;
      m &= m - 1;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// producer wave
// ---------------------------------------------------------------------------------------------------------------------
// ROLL: the rolling cell set-up (ug_k0_gather_quad_roll: 4 registers per item IN FLIGHT instead of 4 per item of the pass) -- what
// lets F >= 4 (P >= 9 levels: 72 set-up registers otherwise) into the 168-register budget of the 12-wave geometry
template <int F, int NBL, int SLOTS, bool ROLL>
__device__ __forceinline__ void ug_pc_producer(const ug_shade_args &a, const float *__restrict__ k0b, const ug_ws_view &ws,
                                               float *__restrict__ rgb_marched, int32_t *__restrict__ tile_counter,
                                               float *ring, unsigned ctl) {
  const int lane = ug_lane();
  const int qs = lane >> 2, qg = lane & 3;
  const ug_quad_axis qa = ug_quad_axis_of(a, qg);
  const int comp = qg < 2 ? qg : 2;
  int seq = 0;            // passes published so far
  int tail_seen = 0;      // last value read from the consumer's counter
  int victim = 0;
  for (;;) {
    const int64_t tile = ug_next_tile(tile_counter, ws.n_tiles, blockIdx.x & 7, victim);
    if (tile < 0) break;
    const int count = ws.count[tile];
    if (count <= 0) {       // nothing survived in this tile: its rays are black, no hand-off needed
      const int64_t ray = tile * UG_WAVE + lane;
      if (ray < a.n_rays) { rgb_marched[3 * ray] = 0.f; rgb_marched[3 * ray + 1] = 0.f; rgb_marched[3 * ray + 2] = 0.f; }
      continue;
    }
    const float *__restrict__ ef = (const float *)(ws.ent + tile * ws.cap);
    const uint8_t *__restrict__ slot = ws.slot + tile * ws.cap;
    // inputs of the next pass are fetched one pass ahead.  UNCONDITIONAL loads on clamped indices (entries past the end of
    // the list repeat its last survivor; the consumer masks them with `ok`): with the loads under exec branches hipcc cannot
    // count them and waits vmcnt(0) at the top of every pass -- for the prefetch it has just issued
    float w_n, pg0_n, pg1_n;
    int sl_n;
    {
      const int e0 = min(lane & 31, count - 1), q0 = min(qs, count - 1), q1 = min(16 + qs, count - 1);
      sl_n = slot[e0]; w_n = ef[4 * e0 + 3];
      pg0_n = ef[4 * q0 + comp]; pg1_n = ef[4 * q1 + comp];
    }
    for (int base = 0; base < count; base += 32) {
      const float ww = w_n, pg0 = pg0_n, pg1 = pg1_n;
      const int sl = sl_n;
      {
        const int e2 = min(base + 32 + (lane & 31), count - 1), q0 = min(base + 32 + qs, count - 1), q1 = min(base + 48 + qs, count - 1);
        sl_n = slot[e2]; w_n = ef[4 * e2 + 3];
        pg0_n = ef[4 * q0 + comp]; pg1_n = ef[4 * q1 + comp];
      }
      float f3[2][3];
      {
        const float pgs[2] = {pg0, pg1};
#ifdef UG_ABL_NO_GATHER
        f3[0][0] = pg0; f3[0][1] = pg1; f3[0][2] = ww; f3[1][0] = pg1; f3[1][1] = pg0; f3[1][2] = ww;
#else
        if constexpr (ROLL) {
          ug_k0_gather_quad_roll<F, NBL, 2>(k0b, a, qa, pgs, f3);
        } else {
          ug_gather_state<F, NBL, 2> gst;
          ug_k0_gather_begin<F, NBL, 2>(k0b, a, qa, pgs, gst);
          ug_k0_gather_finish<F, NBL, 2>(k0b, a, gst, f3);
        }
#endif
      }
      // a free slot: the consumer has taken pass seq - SLOTS (waited for AFTER the gather: the features sit in registers)
      while (seq - tail_seen >= SLOTS) {
        tail_seen = ug_lds_peek(ctl + 4);
        if (seq - tail_seen >= SLOTS) __builtin_amdgcn_s_sleep(UG_PC_PRODUCER_SLEEP);
      }
      float *sp = ring + (seq % SLOTS) * UG_PC_SLOT_FLOATS;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        float *fp = sp + UG_PC_FEAT + (16 * it + qs) * 12 + 3 * qg;
        fp[0] = f3[it][0]; fp[1] = f3[it][1]; fp[2] = f3[it][2];
      }
      if (lane < 32) { sp[UG_PC_W + lane] = ww; ((int *)sp)[UG_PC_SL + lane] = sl; }
      if (lane == 0) { ((int *)sp)[UG_PC_HDR] = (int)tile; ((int *)sp)[UG_PC_HDR + 1] = count; ((int *)sp)[UG_PC_HDR + 2] = base; }
      ++seq;
      ug_lds_publish(ctl, seq);
    }
  }
  // end marker
  while (seq - tail_seen >= SLOTS) {
    tail_seen = ug_lds_peek(ctl + 4);
    if (seq - tail_seen >= SLOTS) __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) ((int *)(ring + (seq % SLOTS) * UG_PC_SLOT_FLOATS))[UG_PC_HDR] = -1;
  ++seq;
  ug_lds_publish(ctl, seq);
}

// ---------------------------------------------------------------------------------------------------------------------
// consumer wave
// ---------------------------------------------------------------------------------------------------------------------
// MODE: 0 = hand-scheduled 4-tile pass (ug_rgbnet_pass_h2), 1 = lean pass.  (A third mode -- two passes of a tile per
// consumer wave in lock step, every weight fragment feeding two MFMAs, layer 3 of a tile behind the next tile's MFMAs, 251
// VGPRs -- was built, verified bit-identical and measured at 4.53-4.56 ms against 4.39 ms: profiles/r03/shade_dual_pass_ab.txt.)
// (Geometries with more consumers than producers -- 4 + 8, 5 + 7, the embedding table in global memory -- were built, verified
// bit-identical and measured 3-5 % slower in round 4: profiles/r04/shade_geometry_48_ab.txt, tools/experiments/ARMS.md.)
template <int PE, int SLOTS, int MODE>
__device__ __forceinline__ void ug_pc_consumer(const ug_shade_args &a, const float *__restrict__ viewdirs, const ug_mlp_lds &M,
                                               float *__restrict__ rgb_marched, const float *ring, unsigned ctl, float *scr) {
  constexpr int C = 12, CH = UG_CH(C), NEMB = 3 + 6 * PE, KL = (2 * CH + NEMB + 1) / 2, EH = KL - CH;
  const int lane = ug_lane();
  const int h = lane >> 5, sv = lane & 31;
  unsigned *amask = (unsigned *)scr;
  float4 *aval = (float4 *)(scr + 64);
  float *embt = scr + UG_ACC_SCRATCH_FLOATS;   // [64 rays][2 halves][EH]
  float accr = 0.f, accg = 0.f, accb = 0.f;    // lane = ray slot of the current tile
  int cur_tile = -1;
  int seq = 0, head_seen = 0;
  ug_h2_state h2st;
  if constexpr (MODE == 0) ug_h2_preload(M, h * 64, h2st);
  amask[lane] = 0u;                // the hand-scheduled passes keep the per-ray masks cleared between passes
  ug_wave_lds_sync();
  for (;;) {
    while (head_seen <= seq) {
      head_seen = ug_lds_peek(ctl);
      if (head_seen <= seq) __builtin_amdgcn_s_sleep(UG_PC_CONSUMER_SLEEP);
    }
    const float *sp = ring + (seq % SLOTS) * UG_PC_SLOT_FLOATS;
    const int tile = __builtin_amdgcn_readfirstlane(((const int *)sp)[UG_PC_HDR]);
    if (tile < 0) break;
    const int count = __builtin_amdgcn_readfirstlane(((const int *)sp)[UG_PC_HDR + 1]);
    const int base = __builtin_amdgcn_readfirstlane(((const int *)sp)[UG_PC_HDR + 2]);
    float x[KL];
    {
      const float *rp = sp + UG_PC_FEAT + sv * 12 + h * 6;
#pragma unroll
      for (int k = 0; k < CH; ++k) x[k] = rp[k];
    }
    const float ww = sp[UG_PC_W + sv];
    const int sl = ((const int *)sp)[UG_PC_SL + sv];
    ++seq;
    ug_lds_publish(ctl + 4, seq);     // the slot's values are in registers: hand it back before the rgbnet starts
    const bool ok = base + sv < count;
    if (tile != cur_tile) {
      if (cur_tile >= 0) {
        const int64_t ray = (int64_t)cur_tile * UG_WAVE + lane;
        if (ray < a.n_rays) { rgb_marched[3 * ray] = accr; rgb_marched[3 * ray + 1] = accg; rgb_marched[3 * ray + 2] = accb; }
      }
      accr = accg = accb = 0.f;
      cur_tile = tile;
      // view-direction embedding of the tile's 64 rays, once per tile (lane = ray slot): as in ug_shade_tile
      int64_t ray = (int64_t)tile * UG_WAVE + lane;
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
#ifdef UG_ABL_NO_EMB_KSTEPS
#pragma unroll
    for (int s = CH; s < KL; ++s) x[s] = 0.f;
#else
    {
      const float *er = embt + sl * (2 * EH) + h * EH;
#pragma unroll
      for (int s = CH; s < KL; ++s) x[s] = er[s - CH];
    }
#endif
#ifdef UG_ABL_NO_CONSUME
    accr += x[0] + ww; accg += x[1]; accb += (float)sl + (ok ? 1.f : 0.f);
    continue;
#endif
    if constexpr (MODE == 1) {
      ug_rgbnet_pass_lean<C, PE>(x, ww, sl, ok, M, amask, aval, accr, accg, accb);
    } else {
      ug_rgbnet_pass_h2<C, PE, true, true>(x, ww, sl, ok, M, amask, aval, accr, accg, accb, h2st);
    }
  }
  if (cur_tile >= 0) {
    const int64_t ray = (int64_t)cur_tile * UG_WAVE + lane;
    if (ray < a.n_rays) { rgb_marched[3 * ray] = accr; rgb_marched[3 * ray + 1] = accg; rgb_marched[3 * ray + 2] = accb; }
  }
}
