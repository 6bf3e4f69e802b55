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
// feeding the vector-memory path, so neither competes with a twin for its unit.  The ring (UG_PC_SLOTS slots per pair)
// decouples memory latency from the MFMA chain.  Hand-off = two monotone LDS counters per pair (head: written by the
// producer only, tail: by the consumer only), polled with ds_read + s_sleep; LDS operations of a wave complete in order, a
// counter is published after `s_waitcnt lgkmcnt(0)`.  Results are bit-identical to k_shade_mlp: same gather, same rgbnet,
// same accumulation order (list order inside a tile; a tile is owned by one pair).
#pragma once
#include "ugrid_render.h"

#ifndef UG_PC_SLOTS
#define UG_PC_SLOTS 4
#endif
#ifndef UG_PC_NBL
// gather items (x 6 dwordx4) in flight per producer wave: 6 at F <= 3; the set-up state grows with the level count
// (4 registers per (round, level)), so F = 4 keeps 5 and F = 5 keeps 4 in flight to stay inside 256 VGPRs without scratch
#ifdef UG_PC_NBL_FIXED            // A/B builds
#define UG_PC_NBL(F) UG_PC_NBL_FIXED
#else
#define UG_PC_NBL(F) ((F) <= 3 ? 6 : ((F) == 4 ? 5 : 4))
#endif
#endif
// ring slot (floats): feat [32][12] | w [32] | sl [32] (int) | hdr {tile, count, base, -} (int)
#define UG_PC_FEAT 0
#define UG_PC_W 384
#define UG_PC_SL 416
#define UG_PC_HDR 448
#define UG_PC_SLOT_FLOATS 452
// per pair: ring | ctl {head, tail, -, -}
#define UG_PC_PAIR_FLOATS (UG_PC_SLOTS * UG_PC_SLOT_FLOATS + 4)

template <int PE>
__host__ __device__ static inline int ug_pc_consumer_scratch_floats() {
  return ug_wave_scratch_floats<12, PE, 2>();      // amask | aval | per-tile embedding table
}
template <int PE>
__host__ __device__ static inline int ug_pc_lds_bytes() {
  return (int)sizeof(float) * (ug_mlp_lds_floats<12, PE, 2>() + 4 * UG_PC_PAIR_FLOATS + 4 * ug_pc_consumer_scratch_floats<PE>());
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

#ifdef UG_SHADE_PROF
// instrumented A/B builds only (tools/gpu_shade_pc_prof.py): g_pc_dbg bit 0 = producers skip the k0 loads (features :=
// positions: WRONG RESULTS, shows the consumer-bound time), bit 1 = consumers skip the rgbnet (shows the producer-bound time)
__device__ int g_pc_dbg;
#define UG_PC_DBG(bit) (g_pc_dbg & (bit))
#define UG_PC_T0(t) const unsigned long long t = __builtin_amdgcn_s_memtime();
#define UG_PC_ADD(acc, t) acc += __builtin_amdgcn_s_memtime() - t;
#else
#define UG_PC_DBG(bit) 0
#define UG_PC_T0(t)
#define UG_PC_ADD(acc, t)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// producer wave
// ---------------------------------------------------------------------------------------------------------------------
template <int F>
__device__ __forceinline__ void ug_pc_producer(const ug_shade_args &a, const float *__restrict__ k0b, const ug_ws_view &ws,
                                               float *__restrict__ rgb_marched, int32_t *__restrict__ tile_counter,
                                               float *ring, unsigned ctl, unsigned long long *pstat) {
  constexpr int NBL = UG_PC_NBL(F);
  const int lane = ug_lane();
  const int qs = lane >> 2, qg = lane & 3;
  const ug_quad_axis qa = ug_quad_axis_of(a, qg);
  const int comp = qg < 2 ? qg : 2;
  int seq = 0;            // passes published so far
  int tail_seen = 0;      // last value read from the consumer's counter
  int victim = 0;
#ifdef UG_SHADE_PROF
  unsigned long long t_wait = 0, t_gather = 0, n_pass = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const int dbg_nogather = UG_PC_DBG(1);
#endif
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
      UG_PC_T0(tg)
      float f3[2][3];
#ifdef UG_SHADE_PROF
      if (dbg_nogather) {
        f3[0][0] = pg0; f3[0][1] = pg0 * 0.5f; f3[0][2] = pg0 * 0.25f; f3[1][0] = pg1; f3[1][1] = pg1 * 0.5f; f3[1][2] = pg1 * 0.25f;
      } else
#endif
      {
        ug_gather_state<F, NBL, 2> gst;
        const float pgs[2] = {pg0, pg1};
        ug_k0_gather_begin<F, NBL, 2>(k0b, a, qa, pgs, gst);
        ug_k0_gather_finish<F, NBL, 2>(k0b, a, gst, f3);
      }
      UG_PC_ADD(t_gather, tg)
      // a free slot: the consumer has taken pass seq - SLOTS (waited for AFTER the gather: the features sit in registers)
      UG_PC_T0(tw)
      while (seq - tail_seen >= UG_PC_SLOTS) {
        tail_seen = ug_lds_peek(ctl + 4);
        if (seq - tail_seen >= UG_PC_SLOTS) __builtin_amdgcn_s_sleep(2);
      }
      UG_PC_ADD(t_wait, tw)
      float *sp = ring + (seq % UG_PC_SLOTS) * UG_PC_SLOT_FLOATS;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        float *fp = sp + UG_PC_FEAT + (16 * it + qs) * 12 + 3 * qg;
        fp[0] = f3[it][0]; fp[1] = f3[it][1]; fp[2] = f3[it][2];
      }
      if (lane < 32) { sp[UG_PC_W + lane] = ww; ((int *)sp)[UG_PC_SL + lane] = sl; }
      if (lane == 0) { ((int *)sp)[UG_PC_HDR] = (int)tile; ((int *)sp)[UG_PC_HDR + 1] = count; ((int *)sp)[UG_PC_HDR + 2] = base; }
      ++seq;
      ug_lds_publish(ctl, seq);
#ifdef UG_SHADE_PROF
      ++n_pass;
#endif
    }
  }
  // end marker
  while (seq - tail_seen >= UG_PC_SLOTS) {
    tail_seen = ug_lds_peek(ctl + 4);
    if (seq - tail_seen >= UG_PC_SLOTS) __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) ((int *)(ring + (seq % UG_PC_SLOTS) * UG_PC_SLOT_FLOATS))[UG_PC_HDR] = -1;
  ++seq;
  ug_lds_publish(ctl, seq);
#ifdef UG_SHADE_PROF
  if (lane == 0) { atomicAdd(pstat + 0, t_gather); atomicAdd(pstat + 1, t_wait); atomicAdd(pstat + 2, n_pass);
                   atomicAdd(pstat + 3, __builtin_amdgcn_s_memtime() - t_begin); }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// consumer wave
// ---------------------------------------------------------------------------------------------------------------------
template <int PE>
__device__ __forceinline__ void ug_pc_consumer(const ug_shade_args &a, const float *__restrict__ viewdirs, const ug_mlp_lds &M,
                                               float *__restrict__ rgb_marched, const float *ring, unsigned ctl, float *scr,
                                               unsigned long long *pstat) {
  constexpr int C = 12, CH = UG_CH(C), NEMB = 3 + 6 * PE, KL = (2 * CH + NEMB + 1) / 2, EH = KL - CH;
  const int lane = ug_lane();
  const int h = lane >> 5, sv = lane & 31;
  unsigned *amask = (unsigned *)scr;
  float4 *aval = (float4 *)(scr + 64);
  float *embt = scr + UG_ACC_SCRATCH_FLOATS;   // [64 rays][2 halves][EH]
  float accr = 0.f, accg = 0.f, accb = 0.f;    // lane = ray slot of the current tile
  int cur_tile = -1;
  int seq = 0, head_seen = 0;
#if UG_MLP_H2
  ug_h2_state h2st;
  ug_h2_preload(M, h * 64, h2st);
  amask[lane] = 0u;                // the hand-scheduled pass keeps the per-ray masks cleared between passes
  ug_wave_lds_sync();
#endif
#ifdef UG_SHADE_PROF
  ug_prof prof_unused;          // phases of ug_rgbnet_pass: acc[3] layer 1, [4] layer 2, [5] layer 3 + sigmoid, [6] accumulation
  for (int i_ = 0; i_ < 8; ++i_) prof_unused.acc[i_] = 0;
#else
  ug_prof prof_unused;
#endif
#ifdef UG_SHADE_PROF
  unsigned long long t_wait = 0, t_mlp = 0, t_tile = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const int dbg_nomlp = UG_PC_DBG(2);
#endif
  for (;;) {
    UG_PC_T0(tw)
    while (head_seen <= seq) {
      head_seen = ug_lds_peek(ctl);
      if (head_seen <= seq) __builtin_amdgcn_s_sleep(2);
    }
    UG_PC_ADD(t_wait, tw)
    const float *sp = ring + (seq % UG_PC_SLOTS) * UG_PC_SLOT_FLOATS;
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
    UG_PC_T0(tt_)
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
    {
      const float *er = embt + sl * (2 * EH) + h * EH;
#pragma unroll
      for (int s = CH; s < KL; ++s) x[s] = er[s - CH];
    }
    UG_PC_ADD(t_tile, tt_)
    UG_PC_T0(tm)
#ifdef UG_SHADE_PROF
    prof_unused.t = tm;
    if (dbg_nomlp) { if (ok && h == 0 && sl == lane) { accr += x[0] * ww; accg += x[1] * ww; accb += x[KL - 1] * ww; } } else
#endif
#if UG_MLP_H2
    ug_rgbnet_pass_h2<C, PE, true, true>(x, ww, sl, ok, M, amask, aval, accr, accg, accb, h2st, prof_unused);
#else
    ug_rgbnet_pass<C, PE, 2>(x, ww, sl, ok, M, amask, aval, accr, accg, accb, prof_unused);
#endif
    UG_PC_ADD(t_mlp, tm)
  }
  if (cur_tile >= 0) {
    const int64_t ray = (int64_t)cur_tile * UG_WAVE + lane;
    if (ray < a.n_rays) { rgb_marched[3 * ray] = accr; rgb_marched[3 * ray + 1] = accg; rgb_marched[3 * ray + 2] = accb; }
  }
#ifdef UG_SHADE_PROF
  if (lane == 0) { atomicAdd(pstat + 4, t_mlp); atomicAdd(pstat + 5, t_wait); atomicAdd(pstat + 6, __builtin_amdgcn_s_memtime() - t_begin);
                   atomicAdd(pstat + 7, t_tile);
                   for (int i_ = 3; i_ < 7; ++i_) atomicAdd(pstat + 8 + i_, prof_unused.acc[i_]); }
#endif
}
