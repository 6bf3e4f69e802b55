// libugrid_hip.so -- shade half of the fused render path (rgbnet on MFMA: fp16x2 / bf16x3 / fp32), the
// single-launch variant and the rgbnet packing kernel.
#include "ugrid_render.h"
#include "ugrid_shade_pc.h"

extern "C" int ug_set_march_waves(int w);  // ugrid_march.hip
extern "C" int ug_set_tv_xcd(int m);        // ugrid_ops.hip
extern "C" int ug_set_train_mlp(int m);     // ugrid_train_mlp.hip
#define UG_PC12_NBL 3        // gather items (x 6 dwordx4) in flight per producer wave of the 12-wave geometry (4 spill)
static int g_shade_pc = 2;   // ugrid_tune("shade_pc", 0|1|2): 2 = 12-wave producer / consumer shade kernel where it applies (default),
                             // 1 = its 8-wave form, 0 = the classic one-wave-does-everything kernel -- bit-identical results,
                             // A/B switch for measurements

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
  for (;;) {
    const int64_t tile = ug_next_tile(tile_counter, ws.n_tiles, blockIdx.x & 7, victim);
    if (tile < 0) break;
    ug_shade_tile<F, C, PE, BF>(a, viewdirs, k0b, M, tile, ws.count[tile], ws.ent + tile * ws.cap,
                                ws.slot + tile * ws.cap, scr, rgb_marched);
  }
}

// producer / consumer shade kernels (ugrid_shade_pc.h): C = 12 quad bricks, fp16x2 rgbnet; NPAIR gather waves + NPAIR rgbnet
// waves per workgroup, one workgroup per CU.  <4, 4 slots, 6 in flight, 4-tile pass> = 8 waves of 256 VGPRs;
// <6, 2 slots, 3 in flight, lean pass> = 12 waves of <= 168 VGPRs (ROLL: the producers' rolling cell set-up, what F >= 4 needs there)
template <int F, int PE, int NPAIR, int SLOTS, int NBL, int MODE, bool ROLL = false>
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
  if (wv < NPAIR) {
    ug_pc_producer<F, NBL, SLOTS, ROLL>(a, k0b, ws, rgb_marched, tile_counter, ring, ctl);
  } else {
    float *scr = pairs + NPAIR * UG_PC_PAIR_FLOATS(SLOTS) + pair * ug_pc_consumer_scratch_floats<PE>();
    ug_pc_consumer<PE, SLOTS, MODE>(a, viewdirs, M, rgb_marched, ring, ctl, scr);
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
extern "C" int64_t ugrid_mlp_packed_bytes(int32_t k0_channels, int32_t viewbase_pe) {
  return (int64_t)sizeof(float) * ug_mlp_lay(k0_channels, 3 + 6 * viewbase_pe).total3;
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

extern "C" int ugrid_tune(const char *key, int value) {
  if (!key) return (int)hipErrorInvalidValue;
  if (!strcmp(key, "march_waves")) return ug_set_march_waves(value) ? (int)hipErrorInvalidValue : 0;
  if (!strcmp(key, "tv_xcd")) return ug_set_tv_xcd(value) ? (int)hipErrorInvalidValue : 0;
  if (!strcmp(key, "train_mlp")) return ug_set_train_mlp(value) ? (int)hipErrorInvalidValue : 0;
  if (!strcmp(key, "shade_pc") && value >= 0 && value <= 2) { g_shade_pc = value; return 0; }
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


template <int F, int PE, int NPAIR, int SLOTS, int NBL, int MODE, bool ROLL = false>
static int ug_shade_pc_launch(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                              ug_ws_view ws, float *rgb, int32_t *counter, hipStream_t st) {
  const int lds_bytes = ug_pc_lds_bytes<PE, NPAIR, SLOTS>();
  if (lds_bytes > 160 * 1024) return (int)hipErrorInvalidValue;   // the CU's LDS
  UG_SET_DYN_LDS((k_shade_pc<F, PE, NPAIR, SLOTS, NBL, MODE, ROLL>), lds_bytes);
  UG_ZERO_WORDS(counter, 8, st);
  // persistent, one workgroup per CU: NPAIR producer waves pull tiles, so a workgroup covers >= NPAIR tiles
  int64_t wgs = (ws.n_tiles + NPAIR - 1) / NPAIR;
  if (wgs > 256) wgs = 256;
  wgs = (wgs + 7) / 8 * 8;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_shade_pc<F, PE, NPAIR, SLOTS, NBL, MODE, ROLL>), dim3((unsigned)wgs), dim3(NPAIR * 128), lds_bytes, st,
                     a, viewdirs, k0b, mlp, ws, rgb, counter);
  UG_LAUNCH_CHECK();
  return 0;
}

template <int F, int C, int PE>
static int ug_shade_launch(const ug_shade_args &a, const float *viewdirs, const float *k0b, const float *mlp,
                           ug_ws_view ws, float *rgb, int32_t *counter, int mlp_mode, hipStream_t st) {
  // fp16x2 keeps a per-wave view-embedding table in LDS next to the rgbnet image: with viewbase_pe = 8 that is 14 KB per
  // consumer wave on top of a 99 KB image -- no 8-wave geometry fits the CU's 160 KB.  ugrid_pack_mlp reports bf16x3 as the
  // best mode for such networks and the fp16x2 kernels are not instantiated for them.
  if constexpr (PE > 4) {
    if (mlp_mode == UGRID_MLP_FP16X2) return (int)hipErrorInvalidValue;
  }
  if constexpr (C == 12 && PE <= 4) {
    if constexpr (F <= 3) {
      if (mlp_mode == UGRID_MLP_FP16X2 && g_shade_pc >= 2)
        return ug_shade_pc_launch<F, PE, 6, 2, UG_PC12_NBL, 1>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
    } else {
      // F >= 4 (P >= 9 levels): the producers' set-up state of a whole pass (4 registers per (round, level)) does not fit the
      // 12-wave geometry's 168 VGPRs; the ROLLING set-up (4 registers per item in flight) does.  Round 5, truck_single.py's shape
      // (F = 4, 1080p, S = 668): 7.55 ms against 8.34-8.40 for the 8-wave geometry, bit-identical (profiles/r05/truck_shade_geometry_ab.txt)
      if (mlp_mode == UGRID_MLP_FP16X2 && g_shade_pc >= 2)
        return ug_shade_pc_launch<F, PE, 6, 2, UG_PC12_NBL, 1, true>(a, viewdirs, k0b, mlp, ws, rgb, counter, st);
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
