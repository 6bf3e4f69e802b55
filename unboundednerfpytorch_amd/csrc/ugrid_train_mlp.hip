// The rgbnet of the TRAINING step (FourierGrid_model.py:233-241, :636: nn.Linear(mlp_in, 128) - ReLU - nn.Linear(128, 128) -
// ReLU - nn.Linear(128, 3) on the step's M ~ 1e5 surviving samples) as hand-written fp32-MFMA kernels.
//
// Why: at M = 84k the three layers are 3.6 GFLOP forward and 7.3 backward -- ~70 us of matrix-pipe time -- but through
// torch.addmm / mm / bmm (rocBLAS, hipBLASLt) the 13 library GEMMs and the elementwise kernels between them leave the GPU
// idle for 50-150 us of host time EACH: 1.46 ms of a 4.4 ms step (profiles/r03/train_step_timeline_*.txt).  Three kernel
// shapes cover the network and its derivative; everything is fp32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32
// accumulation -- the arithmetic class of the library SGEMMs, another summation order).
//
//   k_lin    Y[s][c] = act( sum_k X[s][k] * Wm[c][k] + b[c] ) [* (G[s][c] > 0)]         one wave = 32 samples x NT*32 outputs
//            D[m = s][n = c] = sum_k A[m][k] B[k][n]:  A = the lane's own sample row (registers), B = weights from LDS (k-major
//            image, lanes read consecutive floats).  The MFMA's two k-slots are lane halves (lane >> 5):
//            half h owns the k-range [h * kh, (h + 1) * kh), kh = ceil(K / 2) -- a reordering of the reduction that lets a lane
//            stream ONE contiguous piece of its row instead of every other element.
//            Serves the three forward layers (w_in_major = 0: Wm = weight [out][in]) and the two input-gradient products
//            dX = dY . W (w_in_major = 1: the same weight array read as [in][out], no transpose needed), the latter with the
//            ReLU mask of the layer below folded into the store.
//   k_wgrad  dW[c][k] += sum_s dY[s][c] * X[s][k],  db[c] += sum_s dY[s][c]              one wave = 256 samples x 32 c x all k
//            D[m = c][n = k] with the SAMPLES as the reduction: both operands are read in their natural row-major layout
//            (lanes = consecutive features: coalesced); <= 256 slab partials, added in a fixed order by k_wgrad_reduce
//            (deterministic).  Replaces the library GEMM with a 128 x 128 result and an M-long reduction that runs on 16
//            workgroups (185-225 us at M = 84k; ops.SplitKLinear cut it into a batched GEMM + a sum).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugrid_common.h"
#include "ugrid_hip.h"

#define ST(s) ((hipStream_t)(s))

typedef float mlp_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32 operand / result mapping (lane l, h = l >> 5):
//   a: A[m = l & 31][k = h]     b: B[k = h][n = l & 31]     d[i]: D[m = 8 * (i >> 2) + 4 * h + (i & 3)][n = l & 31]
// (operand mapping pinned by tests/test_gpu_ops.py::test_fused_rgbnet_matches_torch_linear_layers)

// KH: k-steps = elements of the lane's half row (compile time; the weight image is zero padded to 2 * KH rows); NT: 32-wide
// output tiles (outputs = NT * 32, padded); VEC: the half rows are read as 16-byte loads (K = 2 * KH, aligned rows)
#ifndef UG_LIN_THREADS
#define UG_LIN_THREADS 1024
#endif
template <int KH, int NT, bool VEC>
__global__ void __launch_bounds__(UG_LIN_THREADS)
k_lin(const float *__restrict__ X, int64_t M, int K, int ldx, const float *__restrict__ W, int ldw, int n_out, int w_in_major,
      const float *__restrict__ bias, int relu, const float *__restrict__ G, int ldg, float *__restrict__ Y, int ldy,
      const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(M, n_dev);                        // (rows on the device: ug_devn, ugrid_common.h)
  extern __shared__ float lds[];                  // weight image Ws[k][c], rows of NP + 1 floats (zero padded)
  constexpr int NP = NT * 32, NPP = NP + 1, K2 = 2 * KH;
  static_assert(KH % 4 == 0, "k-steps come in rounds of four");
  // coalesced reads of the weight array in ITS order; the odd row pitch keeps the transposing writes off one LDS bank.  All of
  // a thread's elements are fetched before the first is stored (one round trip to L2, not one per element).
  {
    constexpr int N_EL = K2 * NP, ITER = (N_EL + UG_LIN_THREADS - 1) / UG_LIN_THREADS;
    float v[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = threadIdx.x + i * UG_LIN_THREADS;
      int k, c;
      if (w_in_major) { k = e / NP; c = e - k * NP; } else { c = e / K2; k = e - c * K2; }
      v[i] = (e < N_EL && k < K && c < n_out) ? (w_in_major ? W[(int64_t)k * ldw + c] : W[(int64_t)c * ldw + k]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int e = threadIdx.x + i * UG_LIN_THREADS;
      int k, c;
      if (w_in_major) { k = e / NP; c = e - k * NP; } else { c = e / K2; k = e - c * K2; }
      if (e < N_EL) lds[k * NPP + c] = v[i];
    }
  }
  __syncthreads();
  const int lane = ug_lane(), h = lane >> 5, col = lane & 31;
  const int64_t n_tiles = (M + 31) >> 5;
  constexpr int WAVES = UG_LIN_THREADS / 64;
  const float *__restrict__ wl = lds + (h * KH) * NPP + col;
  for (int64_t tile = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6); tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
    const int64_t s = tile * 32 + col;
    const bool row_ok = s < M;
    const float *__restrict__ xr = X + (row_ok ? s : 0) * ldx + h * KH;
    const int k_base = h * KH;
    auto load4 = [&](int j0) -> float4 {           // elements j0 .. j0 + 3 of the lane's half row
      if (VEC) return row_ok ? *(const float4 *)(xr + j0) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 v;
      v.x = (row_ok && k_base + j0 + 0 < K) ? xr[j0 + 0] : 0.f;
      v.y = (row_ok && k_base + j0 + 1 < K) ? xr[j0 + 1] : 0.f;
      v.z = (row_ok && k_base + j0 + 2 < K) ? xr[j0 + 2] : 0.f;
      v.w = (row_ok && k_base + j0 + 3 < K) ? xr[j0 + 3] : 0.f;
      return v;
    };
    mlp_f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // D[m = sample][n = output]: the sample rows are the A operand, so that a store instruction writes two runs of 32
    // consecutive outputs (one per lane half).  Four k-steps per round: their 4 * NT weights come from LDS first, the next
    // four row elements are already on their way.
    float4 xv = load4(0), xn = load4(4 < KH ? 4 : 0);
#pragma unroll 1
    for (int j0 = 0; j0 < KH; j0 += 4) {
      const float4 xnn = load4(j0 + 8 < KH ? j0 + 8 : j0);      // two rounds ahead
      float w[4][NT];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int t = 0; t < NT; ++t) w[jj][t] = wl[(j0 + jj) * NPP + 32 * t];
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[jj], w[jj][t], acc[t], 0, 0, 0);
      xv = xn;
      xn = xnn;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = 32 * t + col;
      if (c >= n_out) continue;
      const float bc = bias ? bias[c] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t sr = tile * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
        if (sr >= M) continue;
        float r = acc[t][i];
        if (bias) r = r + bc;
        if (relu) r = fmaxf(r, 0.f);
        if (G && !(G[sr * ldg + c] > 0.f)) r = 0.f;
        Y[sr * ldy + c] = r;
      }
    }
  }
}

// one wave: 32 output rows (features c of dY) x NT*32 columns (features k of X) over the sample chunks of its SLAB (chunk =
// UG_WG_CHUNK samples; slab b owns chunks b, b + n_slabs, ...); the operands of UG_WG_BATCH reduction steps are fetched before
// their MFMAs are issued.  The slab's tile goes to partial[slab] and k_wgrad_reduce adds the slabs in a fixed order: the
// weight gradient is deterministic (two runs give the same bits), unlike a sum of atomics.
#define UG_WG_CHUNK 128
#define UG_WG_BATCH 8
#define UG_WG_MAX_SLABS 256
template <int NT>
__global__ void __launch_bounds__(256)
k_wgrad(const float *__restrict__ dY, int ldd, int n_out, const float *__restrict__ X, int ldx, int K, int64_t M,
        float *__restrict__ partial_w, float *__restrict__ partial_b, int mt_count, int n_slabs, const int64_t *__restrict__ n_dev) {
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int mt = (int)(wave % mt_count);
  const int slab = (int)(wave / mt_count);
  if (slab >= n_slabs) return;
  UG_DEVN_CLAMP(M, n_dev);
  const int lane = ug_lane(), h = lane >> 5, col = lane & 31;
  const int c = 32 * mt + col;
  const bool c_ok = c < n_out;
  mlp_f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  bool k_ok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) k_ok[t] = 32 * t + col < K;
  for (int64_t s0 = (int64_t)slab * UG_WG_CHUNK; s0 < M; s0 += (int64_t)n_slabs * UG_WG_CHUNK) {
    const int64_t s1 = (s0 + UG_WG_CHUNK < M) ? s0 + UG_WG_CHUNK : M;
    for (int64_t sb = s0; sb < s1; sb += 2 * UG_WG_BATCH) {      // rows >= s1 contribute zeros
      float a[UG_WG_BATCH], b[UG_WG_BATCH][NT];
#pragma unroll
      for (int u = 0; u < UG_WG_BATCH; ++u) {
        const int64_t s = sb + 2 * u + h;
        const bool on = s < s1;
        a[u] = (on && c_ok) ? dY[s * ldd + c] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) b[u][t] = (on && k_ok[t]) ? X[s * ldx + 32 * t + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UG_WG_BATCH; ++u) {
        bsum += a[u];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
      }
    }
  }
  // D[m = 8 g + 4 h + e][n = col]: row = output feature, column = input feature
  float *__restrict__ pw = partial_w + (int64_t)slab * n_out * K;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = 32 * t + col;
    if (k >= K) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int cc = 32 * mt + 8 * (i >> 2) + 4 * h + (i & 3);
      if (cc < n_out) pw[(int64_t)cc * K + k] = acc[t][i];
    }
  }
  if (partial_b) {
    bsum += __shfl_xor(bsum, 32, UG_WAVE);
    if (h == 0 && c_ok) partial_b[(int64_t)slab * n_out + c] = bsum;
  }
}

// out[e] = sum over slabs of partial[slab][e] in a FIXED two-level order: the slabs are cut into 16 consecutive groups, a thread
// adds its group's slabs in ascending order (16 independent loads per round), then the 16 group sums are added in ascending order.
// A workgroup = 16 elements x 16 groups: ~1000 workgroups for a 128 x 128 tile (one thread per element and a serial walk over 256
// slabs left one wave per CU waiting out 16 dependent rounds of loads: 18 us per layer, three times per backward).
// Weights and biases of a layer in ONE launch: elements [0, n_w) are the weight tile, [n_w, n_w + n_b) the bias -- each with its
// own partial array and stride.
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float *__restrict__ partial_w, const float *__restrict__ partial_b, int n_slabs, int n_w, int n_b,
               float *__restrict__ out_w, float *__restrict__ out_b) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  int e = blockIdx.x * 16 + el;
  const bool on = e < n_w + n_b;
  const bool is_b = e >= n_w;
  const float *__restrict__ partial = is_b ? partial_b : partial_w;
  float *__restrict__ out = is_b ? out_b : out_w;
  const int n = is_b ? n_b : n_w;
  if (is_b) e -= n_w;
  const int per = (n_slabs + 15) >> 4, s_lo = grp * per, s_hi = s_lo + per < n_slabs ? s_lo + per : n_slabs;
  float acc = 0.f;
  if (on) {
    for (int b0 = s_lo; b0 < s_hi; b0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (b0 + u < s_hi) ? partial[(int64_t)(b0 + u) * n + e] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += v[u];
    }
  }
  red[grp][el] = acc;
  __syncthreads();
  if (grp == 0 && on) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][el];
    out[e] = t;
  }
}

// ----------------------------------------------------------------------------------------------
// bf16x3 twins of k_lin / k_wgrad (round 6; ugrid_tune("train_mlp", 1), the default).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate: the 128-wide layers of a DVGO step (M ~ 1.2e5) are 25 us of matrix-pipe
// time per launch, and the fp32 kernels reach a third of that rate (profiles/r05/voxgo_train_dvgo_pmc.txt: matrix pipe busy 0.34).
// Every fp32 operand is split into three bf16 parts x = h + m + l (|x - (h + m + l)| <= 2^-24 |x|: the split the render side's
// BF16X3 rgbnet uses, csrc/ugrid_render.h ug_split8) and a product keeps the six part products above 2^-24 of |a||b|
// (m m, l h, h l, m h, h m, h h -- smallest first), accumulated in fp32 by v_mfma_f32_32x32x16_bf16: 6 MFMAs of 32 cycles per 16
// reduction steps against 8 of 64 -- 2.7 x less matrix time at fp32 accuracy, no range guard needed (bf16 has fp32's exponent).
// Operand mapping of v_mfma_f32_32x32x16_bf16 (lane l, h = l >> 5):  a: A[m = l & 31][k = 8 h + e],  b: B[k = 8 h + e][n = l & 31],
// e = 0..7;  d[i]: D[m = 8 (i >> 2) + 4 h + (i & 3)][n = l & 31].
// ----------------------------------------------------------------------------------------------
typedef __bf16 mlp_bf16x8 __attribute__((ext_vector_type(8)));
struct mlp_split3 { mlp_bf16x8 h, m, l; };
__device__ __forceinline__ mlp_split3 mlp_split8(const float (&x)[8]) {
  mlp_split3 s;
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
// (the pinned issue order + scheduling barrier of the render side's bf16 chain: an MFMA never reads as SrcC the accumulator the MFMA issued
// right before it wrote -- csrc/ugrid_render.h UG_MFMA_BF16)
#define MLP_MFMA_BF16(acc, a, b)                                      \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);  \
  __builtin_amdgcn_sched_barrier(0)

// one k-step of a 32-sample x 128-output tile: the sample's 8 inputs (B operand) against the three weight parts of 4 output tiles
// (A operands from the LDS image, wl = image + h * 128 + col), six products per tile, smallest first; consecutive MFMAs never share an
// accumulator
__device__ __forceinline__ void mlp_b3_kstep(mlp_f32x16 (&acc)[4], const mlp_bf16x8 *__restrict__ wl, int ks, const float (&xv)[8]) {
  const mlp_split3 xs = mlp_split8(xv);
  const mlp_bf16x8 *wp = wl + (ks * 3) * 2 * 128;
  mlp_bf16x8 wm[4], wlo[4], wh[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wm[t] = wp[(1 * 2) * 128 + 32 * t];
#pragma unroll
  for (int t = 0; t < 4; ++t) wlo[t] = wp[(2 * 2) * 128 + 32 * t];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wm[t], xs.m); }
#pragma unroll
  for (int t = 0; t < 4; ++t) wh[t] = wp[(0 * 2) * 128 + 32 * t];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wlo[t], xs.h); }
#pragma unroll
  for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wh[t], xs.l); }
#pragma unroll
  for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wm[t], xs.h); }
#pragma unroll
  for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wh[t], xs.m); }
#pragma unroll
  for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wh[t], xs.h); }
}

// the weight image of k_lin_b3 / k_lin_b3_dense: rec[((ks * 3 + part) * 2 + h) * 128 + n] = parts of W[n][16 ks + 8 h + e], e = 0..7
// (zero beyond K / n_out); unconditional loads on clamped indices, zeros by a mask product (see k_wgrad_b3)
template <int KS, int THREADS>
__device__ __forceinline__ void mlp_b3_stage_weights(mlp_bf16x8 *img, const float *__restrict__ W, int ldw, int K, int n_out, int w_in_major) {
  constexpr int N_REC = 2 * KS * 128;
  for (int r = threadIdx.x; r < N_REC; r += THREADS) {
    int n, kq;                                     // kq = 2 ks + h: the record's first input is 8 kq
    if (w_in_major) { n = r & 127; kq = r >> 7; } else { kq = r % (2 * KS); n = r / (2 * KS); }
    const int k0 = 8 * kq, ks = kq >> 1, hh = kq & 1;
    float v[8];
    const int nc = n < n_out ? n : n_out - 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kc = k0 + e < K ? k0 + e : K - 1;
      v[e] = (w_in_major ? W[(int64_t)kc * ldw + nc] : W[(int64_t)nc * ldw + kc]) * ((k0 + e < K && n < n_out) ? 1.f : 0.f);
    }
    const mlp_split3 sp = mlp_split8(v);
    img[((ks * 3 + 0) * 2 + hh) * 128 + n] = sp.h;
    img[((ks * 3 + 1) * 2 + hh) * 128 + n] = sp.m;
    img[((ks * 3 + 2) * 2 + hh) * 128 + n] = sp.l;
  }
}

// bias, ReLU, the ReLU mask of the layer below, 16 float4 stores: D[m = output][n = sample] of one tile -> Y[s][..]
__device__ __forceinline__ void mlp_b3_store(const mlp_f32x16 (&acc)[4], int64_t s, int h, int n_out, const float *__restrict__ bias, int relu,
                                             const float *__restrict__ G, int ldg, float *__restrict__ Y, int ldy) {
  const int nt_live = (n_out + 31) >> 5;          // output tiles that hold anything (wave-uniform)
#pragma unroll
  for (int t0 = 0; t0 < 4; t0 += 2) {              // two output tiles at a time
    if (t0 >= nt_live) continue;
    // the mask's 8 pieces of the pair are requested before the first is used (one exposed wait each otherwise)
    float4 gm[2][4];
    if (G) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = 32 * (t0 + t) + 8 * q + 4 * h;
          gm[t][q] = *(const float4 *)(G + s * ldg + (c0 < n_out ? c0 : n_out - 4));
        }
    }
#pragma unroll
    for (int t = t0; t < t0 + 2; ++t) {
      if (t >= nt_live) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * t + 8 * q + 4 * h;     // outputs c0 .. c0 + 3 of sample s (n_out % 4 == 0: all four or none)
        if (c0 >= n_out) continue;
        float r[4] = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        if (bias) {
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = r[j] + bias[c0 + j];
        }
        if (relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) r[j] = fmaxf(r[j], 0.f);
        }
        if (G) {
          const float4 g = gm[t - t0][q];
          if (!(g.x > 0.f)) r[0] = 0.f;
          if (!(g.y > 0.f)) r[1] = 0.f;
          if (!(g.z > 0.f)) r[2] = 0.f;
          if (!(g.w > 0.f)) r[3] = 0.f;
        }
        *(float4 *)(Y + s * ldy + c0) = make_float4(r[0], r[1], r[2], r[3]);
      }
    }
  }
}

// Y^T tile = W . X^T: D[m = output][n = sample] -- the weights are the A operands (LDS image, 16-byte records of 8 bf16), the sample's
// inputs the B operand, and a lane ends up with FOUR CONSECUTIVE outputs of its sample per accumulator quad: 16 float4 stores per tile
// and lane instead of 64 scalar ones.  KS = k-steps of 16 inputs (zero padded); in k-step ks lane half h multiplies inputs
// 16 ks + 8 h .. + 7 of its sample.  n_out in (32, 128], a multiple of 4.
// The sample rows reach their lanes THROUGH LDS: a lane reading its own row makes every load instruction touch 64 different cache
// lines (66 tag look-ups per instruction, profiles/r06/klin_b3_pmc.txt).
//   k_lin_b3<KS>        K = 16 KS, 16-byte aligned rows: a k-step's 32 x 64 bytes are fetched by two coalesced dwordx4 per lane (four
//                       lanes per 64-byte piece), four k-steps ahead, written to the wave's staging tile (rows padded to 80 bytes: the
//                       transposed reads are conflict-free) and read back in operand order;
//   k_lin_b3_dense<KS>  any K <= 16 KS <= 48 with ldx == K: the tile's 32 rows are ONE contiguous run of 32 K floats, fetched whole by
//                       coalesced dword loads (the next tile's before this tile's MFMAs), rows padded to 16 KS + 1 floats in LDS.
// Every load is unconditional on a clamped address -- rows past the end repeat the last row and feed output columns that are never
// stored: with loads under exec branches hipcc cannot count them and waits vmcnt(0) right behind the look-ahead it has just issued.
#define UG_LINB_THREADS 512
#define UG_LINB_XT_FLOATS (2 * 32 * 20)      /* k_lin_b3, per wave: two staging tiles of 32 rows x 20 floats */
template <int KS>
__global__ void __launch_bounds__(UG_LINB_THREADS)
k_lin_b3(const float *__restrict__ X, int64_t M, int K, int ldx, const float *__restrict__ W, int ldw, int n_out, int w_in_major,
         const float *__restrict__ bias, int relu, const float *__restrict__ G, int ldg, float *__restrict__ Y, int ldy,
         const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(M, n_dev);
  if (M <= 0) return;                              // (a device count of zero)
  extern __shared__ float lds[];
  mlp_bf16x8 *img = (mlp_bf16x8 *)lds;
  const int lane = ug_lane(), h = lane >> 5, col = lane & 31;
  const int64_t n_tiles = (M + 31) >> 5;
  constexpr int WAVES = UG_LINB_THREADS / 64;
  float *xt = lds + 2 * KS * 128 * 3 * 4 + (threadIdx.x >> 6) * UG_LINB_XT_FLOATS;      // this wave's staging tiles
  struct f8 { float v[8]; };
  // per k-step a lane fetches 16 bytes of row (lane >> 2) and of row 16 + (lane >> 2) of the tile, piece lane & 3
  struct rowctx { const float *p0, *p1; };
  auto ctx_of = [&](int64_t tile) -> rowctx {
    rowctx r;
    const int64_t last = M - 1, r0 = tile * 32 + (lane >> 2), r1 = r0 + 16;
    r.p0 = X + (r0 < last ? r0 : last) * ldx + 4 * (lane & 3);
    r.p1 = X + (r1 < last ? r1 : last) * ldx + 4 * (lane & 3);
    return r;
  };
  auto load8 = [&](const rowctx &rc, int ks) -> f8 {
    f8 r;
    const float4 a = *(const float4 *)(rc.p0 + 16 * ks);
    const float4 b = *(const float4 *)(rc.p1 + 16 * ks);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
  };
  constexpr int AHEAD = KS < 2 ? KS : 2;           // k-steps of rows in flight
  // the first tile's rows are requested BEFORE the weights are staged (an HBM round trip beside the 96 KB of split + LDS writes),
  // every later tile's before the previous tile's stores
  int64_t tile = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  rowctx rc = ctx_of(tile);
  f8 xq[AHEAD];
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) xq[a] = load8(rc, a);
  mlp_b3_stage_weights<KS, UG_LINB_THREADS>(img, W, ldw, K, n_out, w_in_major);
  __syncthreads();
  const mlp_bf16x8 *__restrict__ wl = img + h * 128 + col;
  // The k-step loop is software-pipelined by hand (an in-order wave only overlaps what is issued BEHIND an MFMA that occupies the pipe):
  //   * the weight parts live in three register sets that are reloaded IN PLACE right after their last use, for the next k-step
  //     (circularly: after the tile's last k-step comes k-step 0 of the next tile -- the same weights);
  //   * the next k-step's inputs go through the staging tile (write, wave sync, transposed read) and are split into their bf16 parts
  //     between this k-step's MFMA groups.
  // Same products in the same order per accumulator as mlp_b3_kstep: the results do not change.
  auto wpart = [&](int ks, int part, mlp_bf16x8 (&w)[4]) {
    const mlp_bf16x8 *wp = wl + ((ks % KS) * 3 + part) * 2 * 128;
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = wp[32 * t];
  };
  auto stage_write = [&](int ks, const f8 &xv) {
    float *buf = xt + (ks & 1) * (32 * 20);
    *(float4 *)(buf + (lane >> 2) * 20 + 4 * (lane & 3)) = make_float4(xv.v[0], xv.v[1], xv.v[2], xv.v[3]);
    *(float4 *)(buf + (16 + (lane >> 2)) * 20 + 4 * (lane & 3)) = make_float4(xv.v[4], xv.v[5], xv.v[6], xv.v[7]);
  };
  auto stage_read = [&](int ks, float (&own)[8]) {
    const float *buf = xt + (ks & 1) * (32 * 20) + col * 20 + 8 * h;
    const float4 a = *(const float4 *)buf, b = *(const float4 *)(buf + 4);
    own[0] = a.x; own[1] = a.y; own[2] = a.z; own[3] = a.w; own[4] = b.x; own[5] = b.y; own[6] = b.z; own[7] = b.w;
  };
  mlp_bf16x8 wh[4], wm[4], wlo[4];                 // parts 0, 1, 2 of the current k-step
  wpart(0, 1, wm); wpart(0, 2, wlo); wpart(0, 0, wh);
  for (; tile < n_tiles; ) {
    const int64_t s = tile * 32 + col;
    mlp_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // k-step 0's inputs: through the staging tile, not overlapped (once per tile)
    mlp_split3 xs;
    {
      stage_write(0, xq[0]);
      if (AHEAD < KS) xq[0] = load8(rc, AHEAD);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      float own[8];
      stage_read(0, own);
      xs = mlp_split8(own);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bool more = ks + 1 < KS;
      float own_n[8];
      mlp_split3 xn;
      if (more) {                                  // the next k-step's coalesced pieces -> staging tile (no wait yet)
        stage_write(ks + 1, xq[(ks + 1) % AHEAD]);
        if (ks + 1 + AHEAD < KS) xq[(ks + 1) % AHEAD] = load8(rc, ks + 1 + AHEAD);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wm[t], xs.m); }
      if (more) {                                  // the writes have landed behind four MFMAs: transposed read of the wave's own rows
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        stage_read(ks + 1, own_n);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wlo[t], xs.h); }
      wpart(ks + 1, 2, wlo);                       // part l: last use was the group above
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wh[t], xs.l); }
      if (more) xn = mlp_split8(own_n);            // (VALU behind the MFMAs)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wm[t], xs.h); }
      wpart(ks + 1, 1, wm);                        // part m: last use was the group above
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wh[t], xs.m); }
#pragma unroll
      for (int t = 0; t < 4; ++t) { MLP_MFMA_BF16(acc[t], wh[t], xs.h); }
      wpart(ks + 1, 0, wh);                        // part h
      __builtin_amdgcn_sched_barrier(0);
      if (more) xs = xn;
    }
    tile += (int64_t)gridDim.x * WAVES;
    rc = ctx_of(tile);
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) xq[a] = load8(rc, a);      // (past the last tile: the last row again, never used)
    if (s < M) mlp_b3_store(acc, s, h, n_out, bias, relu, G, ldg, Y, ldy);
  }
}

template <int KS>
__global__ void __launch_bounds__(UG_LINB_THREADS)
k_lin_b3_dense(const float *__restrict__ X, int64_t M, int K, const float *__restrict__ W, int ldw, int n_out, int w_in_major,
               const float *__restrict__ bias, int relu, const float *__restrict__ G, int ldg, float *__restrict__ Y, int ldy,
               const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(M, n_dev);
  if (M <= 0) return;
  extern __shared__ float lds[];
  mlp_bf16x8 *img = (mlp_bf16x8 *)lds;
  constexpr int RP = 16 * KS + 1;                  // staged row pitch in floats: odd, so that the 32 lanes of a transposed read hit 32 banks
  constexpr int NL = (32 * 16 * KS + 63) / 64;     // dwords per lane that cover a tile of 32 rows x K <= 16 KS floats
  const int lane = ug_lane(), h = lane >> 5, col = lane & 31;
  const int64_t n_tiles = (M + 31) >> 5;
  constexpr int WAVES = UG_LINB_THREADS / 64;
  float *xt = lds + 2 * KS * 128 * 3 * 4 + (threadIdx.x >> 6) * (32 * RP + 4);     // (+ a dump slot for the elements past row 31)
  // where element lane + 64 j of a tile's contiguous run lands in the staging tile (the same for every tile)
  int off[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int i = lane + 64 * j, row = i / K;
    off[j] = row < 32 ? row * RP + (i - row * K) : 32 * RP;      // (past the tile: the dump slot -- unconditional writes, no lane masks to keep)
  }
  for (int i = lane; i < 32 * RP; i += 64) xt[i] = 0.f;      // the pad columns K .. 16 KS - 1 stay zero for the whole kernel
  const int64_t total = M * K;
  auto load_tile = [&](int64_t tile, float (&v)[NL]) {
    const int64_t base = tile * 32 * K, left = total - base;            // (tile < n_tiles: left >= 1)
    const int last = left > (int64_t)(64 * NL) ? 64 * NL - 1 : (int)left - 1;      // 32-bit clamp: no 64-bit compare (= an SGPR pair) per load
    const float *__restrict__ xb = X + base;
#pragma unroll
    for (int j = 0; j < NL; ++j) v[j] = xb[min(lane + 64 * j, last)];
  };
  int64_t tile = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  float cur[NL];
  load_tile(tile < n_tiles ? tile : n_tiles - 1, cur);
  mlp_b3_stage_weights<KS, UG_LINB_THREADS>(img, W, ldw, K, n_out, w_in_major);
  __syncthreads();
  const mlp_bf16x8 *__restrict__ wl = img + h * 128 + col;
  for (; tile < n_tiles; ) {
    const int64_t s = tile * 32 + col;
#pragma unroll
    for (int j = 0; j < NL; ++j) xt[off[j]] = cur[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    tile += (int64_t)gridDim.x * WAVES;
    load_tile(tile < n_tiles ? tile : n_tiles - 1, cur);      // the next tile's run, under this tile's MFMAs
    mlp_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const float *rp = xt + col * RP + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float own[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) own[e] = rp[16 * ks + e];
      mlp_b3_kstep(acc, wl, ks, own);
    }
    __builtin_amdgcn_wave_barrier();               // (the next tile's writes follow this tile's reads in the wave's own LDS order)
    if (s < M) mlp_b3_store(acc, s, h, n_out, bias, relu, G, ldg, Y, ldy);
  }
}

// dW[c][k] partials with the SAMPLES as the reduction, bf16x3: one wave = 32 output features c x 64 input features k (two column
// tiles) over the sample chunks of its slab -- A[m = c][kk = sample 8 h + e] = dY, B[kk][n = k] = X, 16 samples per MFMA group.
// Eight dword loads per operand column and step (lanes = consecutive features: coalesced), fetched four steps ahead.
template <int NT>      // column tiles per wave (>= 2: see the MFMA order below)
__global__ void __launch_bounds__(256)
k_wgrad_b3(const float *__restrict__ dY, int ldd, int n_out, const float *__restrict__ X, int ldx, int K, int64_t M,
           float *__restrict__ partial_w, float *__restrict__ partial_b, int mt_count, int kt_count, int n_slabs,
           const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(M, n_dev);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int mt = (int)(wave % mt_count);
  const int kt = (int)((wave / mt_count) % kt_count);
  const int slab = (int)(wave / ((int64_t)mt_count * kt_count));
  if (slab >= n_slabs) return;
  const int lane = ug_lane(), h = lane >> 5, col = lane & 31;
  const int c = 32 * mt + col;
  const bool c_ok = c < n_out;
  const int kcol0 = 32 * NT * kt + col;
  mlp_f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  bool k_ok[NT];
  int kc[NT];                                      // clamped column of every tile (loads are unconditional)
#pragma unroll
  for (int t = 0; t < NT; ++t) { k_ok[t] = kcol0 + 32 * t < K; kc[t] = k_ok[t] ? kcol0 + 32 * t : K - 1; }
  const int cc = c_ok ? c : n_out - 1;
  const float mc = c_ok ? 1.f : 0.f;
  const bool all_cols = (n_out & 31) == 0 && 32 * NT * (kt + 1) <= K;      // (wave-uniform) every lane's columns exist
  float mk[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) mk[t] = k_ok[t] ? 1.f : 0.f;
  struct opnd { float a[8], b[NT][8]; };
  // The wave's work as ONE sequence of 16-sample steps: step t = rows 16 (t & 7) .. + 15 of chunk (t >> 3) of its slab (chunks slab,
  // slab + n_slabs, ...).  Operands are fetched FOUR steps ahead (a ring of four register sets): with two waves per SIMD a step's
  // split + 12 MFMAs are ~900 cycles, an HBM round trip 2-4 k -- one step of look-ahead left the wave waiting most of the time (59 us at
  // M = 1.2e5, round 6 visit I).
  const int64_t n_chunks = (M + UG_WG_CHUNK - 1) / UG_WG_CHUNK;
  const int64_t T = slab < n_chunks ? ((n_chunks - slab + n_slabs - 1) / n_slabs) * (UG_WG_CHUNK / 16) : 0;
  auto fetch = [&](int64_t t) -> opnd {            // (only called with T > 0, hence M > 0; steps past the end load row M - 1 and select zeros)
    opnd o;
    const bool live = t < T;
    const int64_t sb = ((int64_t)slab + (t >> 3) * n_slabs) * UG_WG_CHUNK + 16 * (t & 7) + 8 * h;
    // Unconditional loads everywhere: hipcc turns `cond ? load : 0` back into a branch around the load, and a load under an exec branch
    // is followed by vmcnt(0) -- each of the 24 loads of a step waited for in turn.  One 64-bit base per operand and step, 32-bit row
    // offsets (a 64-bit multiply-add per load made the kernel VALU-bound).
    if (live && all_cols && sb - 8 * h + 16 <= M) {   // (wave-uniform) the common step: every row and every column exists -- no masks
      const float *__restrict__ pa = dY + sb * ldd + c;
      const float *__restrict__ pb = X + sb * ldx + kcol0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o.a[e] = pa[e * ldd];
#pragma unroll
        for (int q = 0; q < NT; ++q) o.b[q][e] = pb[e * ldx + 32 * q];
      }
    } else {                                          // edges: rows past the end re-read row M - 1, zeros by a mask PRODUCT
      const int64_t sbc = sb < M - 1 ? sb : M - 1;
      const float *__restrict__ pa = dY + sbc * ldd + cc;
      const float *__restrict__ pb = X + sbc * ldx;
      const int room = (int)(M - 1 - sbc);           // rows after sbc that exist (>= 0)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool on = live && sb + e < M;
        const int ec = e <= room ? e : room;
        const float mrow = on ? 1.f : 0.f;
        o.a[e] = pa[ec * ldd] * (mrow * mc);
#pragma unroll
        for (int q = 0; q < NT; ++q) o.b[q][e] = pb[ec * ldx + kc[q]] * (mrow * mk[q]);
      }
    }
    return o;
  };
  auto consume = [&](const opnd &cur) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum += cur.a[e];
    const mlp_split3 as = mlp_split8(cur.a);
    mlp_split3 bs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bs[t] = mlp_split8(cur.b[t]);
    __builtin_amdgcn_sched_barrier(0);
    // products outer, column tiles inner: consecutive MFMAs never share an accumulator (NT >= 2)
#pragma unroll
    for (int t = 0; t < NT; ++t) { MLP_MFMA_BF16(acc[t], as.m, bs[t].m); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { MLP_MFMA_BF16(acc[t], as.l, bs[t].h); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { MLP_MFMA_BF16(acc[t], as.h, bs[t].l); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { MLP_MFMA_BF16(acc[t], as.m, bs[t].h); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { MLP_MFMA_BF16(acc[t], as.h, bs[t].m); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { MLP_MFMA_BF16(acc[t], as.h, bs[t].h); }
  };
  if (T > 0) {
    opnd q0 = fetch(0), q1 = fetch(1), q2 = fetch(2), q3 = fetch(3);
    for (int64_t t = 0; t < T; t += 4) {            // (T is a multiple of 8)
      consume(q0); q0 = fetch(t + 4);
      consume(q1); q1 = fetch(t + 5);
      consume(q2); q2 = fetch(t + 6);
      consume(q3); q3 = fetch(t + 7);
    }
  }
  float *__restrict__ pw = partial_w + (int64_t)slab * n_out * K;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = kcol0 + 32 * t;
    if (k >= K) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int cc = 32 * mt + 8 * (i >> 2) + 4 * h + (i & 3);
      if (cc < n_out) pw[(int64_t)cc * K + k] = acc[t][i];
    }
  }
  if (partial_b && kt == 0) {
    bsum += __shfl_xor(bsum, 32, UG_WAVE);
    if (h == 0 && c_ok) partial_b[(int64_t)slab * n_out + c] = bsum;
  }
}

// Y[s][c] = (G[s][c] > 0) ? sum_{j < K} X[s][j] * W[j][c] : 0 for K <= 4 (the gradient of the 3-channel logits pushed through the
// last layer): three FMAs per element, one float4 of outputs per lane -- no matrix pipe needed
__global__ void __launch_bounds__(256)
k_lin_smallk(const float *__restrict__ X, int64_t M, int K, const float *__restrict__ W, int n_out, const float *__restrict__ G,
             float *__restrict__ Y, const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(M, n_dev);
  const int q4 = n_out >> 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < M * q4; idx += (int64_t)gridDim.x * blockDim.x) {
  const int64_t s = idx / q4;
  const int c = (int)(idx - s * q4) * 4;
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < K; ++j) {
    const float x = X[s * K + j];
    const float4 w = *(const float4 *)(W + (int64_t)j * n_out + c);
    r[0] += x * w.x; r[1] += x * w.y; r[2] += x * w.z; r[3] += x * w.w;
  }
  if (G) {
    const float4 g = *(const float4 *)(G + s * n_out + c);
    if (!(g.x > 0.f)) r[0] = 0.f;
    if (!(g.y > 0.f)) r[1] = 0.f;
    if (!(g.z > 0.f)) r[2] = 0.f;
    if (!(g.w > 0.f)) r[3] = 0.f;
  }
  *(float4 *)(Y + s * n_out + c) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// ugrid_tune("train_mlp", 0 | 1): arithmetic of the 33..128-wide products of the training rgbnet -- 0 = fp32 MFMA (k_lin / k_wgrad),
// 1 = bf16x3 (k_lin_b3 / k_wgrad_b3, default).  Both are fp32-accurate; the results differ in the last bits (another rounding of the products).
static int g_train_mlp = 1;
extern "C" int ug_set_train_mlp(int m) { if (m < 0 || m > 1) return 1; g_train_mlp = m; return 0; }

static inline int ug_lin_launch(const float *X, int64_t M, int K, int ldx, const float *W, int ldw, int n_out, int w_in_major, const float *bias,
                                int relu, const float *G, int ldg, float *Y, int ldy, hipStream_t st) {
  if (M <= 0) return 0;
  if (K < 1 || K > 128 || n_out < 1 || n_out > 128) return (int)hipErrorNotSupported;
  const int kh = (K + 1) / 2, nt = n_out <= 32 ? 1 : 4;
  const int64_t tiles = (ug_launch_rows(M) + 31) / 32;      // (grid only: the kernels loop over the tiles of the rows they find)
  constexpr int WAVES = UG_LIN_THREADS / 64;
  int64_t wgs = (tiles + WAVES - 1) / WAVES;
  if (wgs > 256) wgs = 256;                        // one persistent 16-wave workgroup per CU: the weights are staged once
#define UG_LIN_GO(KH, NT_, VEC_)                                                                                                 \
  {                                                                                                                              \
    constexpr int lds = 2 * KH * (NT_ * 32 + 1) * 4;                                                                             \
    UG_SET_DYN_LDS((k_lin<KH, NT_, VEC_>), lds);   /* per device */                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lin<KH, NT_, VEC_>), dim3((unsigned)wgs), dim3(UG_LIN_THREADS), lds, st, X, M, K, ldx, W, ldw, \
                       n_out, w_in_major, bias, relu, G, ldg, Y, ldy, ug_tl_devn.ptr);                                           \
  }
  const bool vec = K == 128 && (ldx & 3) == 0 && ((uintptr_t)X & 15) == 0;
  if (K <= 4 && w_in_major && !bias && !relu && ldx == K && ldw == n_out && ldy == n_out && (!G || ldg == n_out) && (n_out & 3) == 0 &&
      ((((uintptr_t)W) | ((uintptr_t)Y) | ((uintptr_t)G)) & 15) == 0) {
    hipLaunchKernelGGL(k_lin_smallk, dim3((unsigned)((ug_launch_rows(M) * (n_out >> 2) + 255) / 256)), dim3(256), 0, st, X, M, K, W, n_out, G, Y,
                       ug_tl_devn.ptr);
    UG_LAUNCH_CHECK();
    return 0;
  }
  if (g_train_mlp == 1 && nt == 4 && (n_out & 3) == 0 && (ldy & 3) == 0 && ((uintptr_t)Y & 15) == 0 &&
      (!G || ((ldg & 3) == 0 && ((uintptr_t)G & 15) == 0))) {
    constexpr int WB = UG_LINB_THREADS / 64;
    int64_t wgb = (tiles + WB - 1) / WB;
    if (wgb > 256) wgb = 256;                      // one persistent 8-wave workgroup per CU (the image is 12 KB per k-step)
#define UG_LINB_GO(KS_)                                                                                                          \
  {                                                                                                                              \
    constexpr int lds = KS_ * 2 * 128 * 3 * 16 + (UG_LINB_THREADS / 64) * UG_LINB_XT_FLOATS * 4;                                 \
    UG_SET_DYN_LDS((k_lin_b3<KS_>), lds);                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lin_b3<KS_>), dim3((unsigned)wgb), dim3(UG_LINB_THREADS), lds, st, X, M, K, ldx, W, ldw,  \
                       n_out, w_in_major, bias, relu, G, ldg, Y, ldy, ug_tl_devn.ptr);                                           \
    UG_LAUNCH_CHECK();                                                                                                           \
    return 0;                                                                                                                    \
  }
#define UG_LINB_DENSE(KS_)                                                                                                       \
  {                                                                                                                              \
    constexpr int lds = KS_ * 2 * 128 * 3 * 16 + (UG_LINB_THREADS / 64) * (32 * (16 * KS_ + 1) + 4) * 4;                         \
    UG_SET_DYN_LDS((k_lin_b3_dense<KS_>), lds);                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_lin_b3_dense<KS_>), dim3((unsigned)wgb), dim3(UG_LINB_THREADS), lds, st, X, M, K, W, ldw, \
                       n_out, w_in_major, bias, relu, G, ldg, Y, ldy, ug_tl_devn.ptr);                                           \
    UG_LAUNCH_CHECK();                                                                                                           \
    return 0;                                                                                                                    \
  }
    const bool al = (ldx & 3) == 0 && ((uintptr_t)X & 15) == 0;
    if (al && K == 128) UG_LINB_GO(8)
    if (al && K == 64) UG_LINB_GO(4)
    if (al && K == 32) UG_LINB_GO(2)
    if (ldx == K && K <= 48) {                     // dense rows of any length up to 48 (the rgbnet's first layer: K = C + 3 + 6 pe = 39)
      if (K <= 32) UG_LINB_DENSE(2)
      UG_LINB_DENSE(3)
    }
    // (other shapes -- unaligned rows longer than 48, strided rows -- take the fp32-MFMA kernels below)
#undef UG_LINB_GO
#undef UG_LINB_DENSE
  }
  if (nt == 4) {
    if (kh <= 12) UG_LIN_GO(12, 4, false)
    else if (kh <= 20) UG_LIN_GO(20, 4, false)
    else if (kh <= 24) UG_LIN_GO(24, 4, false)
    else if (kh <= 32) UG_LIN_GO(32, 4, false)
    else if (vec) UG_LIN_GO(64, 4, true)
    else UG_LIN_GO(64, 4, false)
  } else {
    if (kh <= 32) UG_LIN_GO(32, 1, false)
    else if (vec) UG_LIN_GO(64, 1, true)
    else UG_LIN_GO(64, 1, false)
  }
#undef UG_LIN_GO
  UG_LAUNCH_CHECK();
  return 0;
}

// partial: UG_WG_MAX_SLABS * (n_out * K + n_out) floats of scratch
static inline int ug_wgrad_launch(const float *dY, int ldd, int n_out, const float *X, int ldx, int K, int64_t M, float *dW, float *db,
                                  float *partial, hipStream_t st) {
  if (K < 1 || K > 128 || n_out < 1 || n_out > 128) return (int)hipErrorNotSupported;
  if (M <= 0) {
    UG_HIP(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)n_out * K, st));
    if (db) UG_HIP(hipMemsetAsync(db, 0, sizeof(float) * (size_t)n_out, st));
    return 0;
  }
  const int mt_count = (n_out + 31) / 32;
  const int64_t chunks = (M + UG_WG_CHUNK - 1) / UG_WG_CHUNK;
  const int n_slabs = (int)(chunks < UG_WG_MAX_SLABS ? chunks : UG_WG_MAX_SLABS);
  float *pw = partial, *pb = db ? partial + (size_t)UG_WG_MAX_SLABS * n_out * K : nullptr;
  const int64_t waves = (int64_t)n_slabs * mt_count;
  const dim3 gr((unsigned)((waves + 3) / 4)), bl(256);
  if (g_train_mlp == 1 && K > 32) {                // bf16x3: one wave = 32 output features x 64 input features
    const int kt_count = (K + 63) / 64;
    const int64_t wv = (int64_t)n_slabs * mt_count * kt_count;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad_b3<2>), dim3((unsigned)((wv + 3) / 4)), bl, 0, st, dY, ldd, n_out, X, ldx, K, M, pw, pb, mt_count,
                       kt_count, n_slabs, ug_tl_devn.ptr);
  } else if (K <= 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad<1>), gr, bl, 0, st, dY, ldd, n_out, X, ldx, K, M, pw, pb, mt_count, n_slabs, ug_tl_devn.ptr);
  else if (K <= 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad<2>), gr, bl, 0, st, dY, ldd, n_out, X, ldx, K, M, pw, pb, mt_count, n_slabs, ug_tl_devn.ptr);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wgrad<4>), gr, bl, 0, st, dY, ldd, n_out, X, ldx, K, M, pw, pb, mt_count, n_slabs, ug_tl_devn.ptr);
  const int n_w = n_out * K, n_b = db ? n_out : 0;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((n_w + n_b + 15) / 16), dim3(256), 0, st, pw, pb, n_slabs, n_w, n_b, dW, db);
  UG_LAUNCH_CHECK();
  return 0;
}

// ----------------------------------------------------------------------------------------------
// The 3-channel last layer without the matrix pipe.  Its three passes through k_lin / k_wgrad pad the 3 logits to a 32-wide MFMA
// tile (10 x the arithmetic) and read the [M,W] activations h2 three times; as VALU kernels they are streams over h2:
//   k_l3_fwd   logits[s][c] = sum_k h2[s][k] W3[c][k] + b3[c]                              reads h2 once
//   k_l3_bwd   dW3[c][k] = sum_s g[s][c] h2[s][k],  db3[c] = sum_s g[s][c],
//              g_h2[s][k] = h2[s][k] > 0 ? g[s][0] W3[0][k] + g[s][1] W3[1][k] + g[s][2] W3[2][k] : 0   (k_lin_smallk's expression)
//                                                                                          reads h2 once, writes g_h2
// Both need W % 4 == 0 (float4 columns); the weight gradient is a fixed-order sum of per-block partials (k_wgrad_reduce).
// ----------------------------------------------------------------------------------------------
#define UG_L3_MAX_BLOCKS 1024
template <int LPR>      // lanes per row: the power of two >= W / 4
__global__ void __launch_bounds__(256)
k_l3_fwd(const float *__restrict__ h2, int64_t M, int W, const float *__restrict__ w3, const float *__restrict__ b3,
         float *__restrict__ logits, const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(M, n_dev);
  constexpr int RPW = UG_WAVE / LPR;                 // rows per wave and iteration
  const int lane = ug_lane(), sub = lane % LPR, rw = lane / LPR;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const bool col_ok = 4 * sub < W;
  float4 wv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) wv[c] = col_ok ? *(const float4 *)(w3 + (int64_t)c * W + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float bias[3] = {b3[0], b3[1], b3[2]};
  for (int64_t s0 = wave * RPW; s0 < M; s0 += n_waves * RPW) {
    const int64_t s = s0 + rw;
    const bool on = s < M && col_ok;
    const float4 h = on ? *(const float4 *)(h2 + s * W + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      r[c] = h.x * wv[c].x;
      r[c] += h.y * wv[c].y;
      r[c] += h.z * wv[c].z;
      r[c] += h.w * wv[c].w;
    }
#pragma unroll
    for (int o = LPR >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int c = 0; c < 3; ++c) r[c] += __shfl_xor(r[c], o, UG_WAVE);
    if (sub == 0 && s < M) {
#pragma unroll
      for (int c = 0; c < 3; ++c) logits[3 * s + c] = r[c] + bias[c];
    }
  }
}

__global__ void __launch_bounds__(256)
k_l3_bwd(const float *__restrict__ g, const float *__restrict__ h2, int64_t M, int W, const float *__restrict__ w3,
         float *__restrict__ g_h2, float *__restrict__ partial_w, float *__restrict__ partial_b, int rows_per_block,
         const int64_t *__restrict__ n_dev) {
  UG_DEVN_CLAMP(M, n_dev);
  __shared__ float red[3840];                        // [row groups][3 W + 3]: (1024 / W) * (3 W + 3) <= 3840 floats
  const int W4 = W >> 2, RPI = 256 / W4;             // float4 columns per row; rows per iteration
  const int t = threadIdx.x, col4 = t % W4, grp = t / W4;
  const bool live = grp < RPI;
  float4 wv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) wv[c] = *(const float4 *)(w3 + (int64_t)c * W + 4 * col4);
  float acc[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float bs[3] = {0.f, 0.f, 0.f};
  if (live) {
    // the block's row chunks: one when the grid covers the rows (the host-counted case), more when a device count exceeds the hint
    for (int64_t r0 = (int64_t)blockIdx.x * rows_per_block; r0 < M; r0 += (int64_t)gridDim.x * rows_per_block) {
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (int64_t s = r0 + grp; s < r1; s += RPI) {
      const float g0 = g[3 * s], g1 = g[3 * s + 1], g2 = g[3 * s + 2];
      const float4 h = *(const float4 *)(h2 + s * W + 4 * col4);
      const float hv[4] = {h.x, h.y, h.z, h.w};
      const float w0[4] = {wv[0].x, wv[0].y, wv[0].z, wv[0].w}, w1[4] = {wv[1].x, wv[1].y, wv[1].z, wv[1].w},
                  w2[4] = {wv[2].x, wv[2].y, wv[2].z, wv[2].w};
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[0][q] += g0 * hv[q];
        acc[1][q] += g1 * hv[q];
        acc[2][q] += g2 * hv[q];
        float r = 0.f;
        r += g0 * w0[q];
        r += g1 * w1[q];
        r += g2 * w2[q];
        o[q] = hv[q] > 0.f ? r : 0.f;
      }
      *(float4 *)(g_h2 + s * W + 4 * col4) = make_float4(o[0], o[1], o[2], o[3]);
      if (col4 == 0) {
        bs[0] += g0;
        bs[1] += g1;
        bs[2] += g2;
      }
    }
    }
    const int stride = 3 * W + 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[grp * stride + c * W + 4 * col4 + q] = acc[c][q];
      if (col4 == 0) red[grp * stride + 3 * W + c] = bs[c];
    }
  }
  __syncthreads();
  const int stride = 3 * W + 3;
  for (int e = t; e < stride; e += 256) {
    float v = 0.f;
    for (int gq = 0; gq < RPI; ++gq) v += red[gq * stride + e];      // fixed order over the row groups
    if (e < 3 * W) partial_w[(int64_t)blockIdx.x * 3 * W + e] = v;
    else partial_b[(int64_t)blockIdx.x * 3 + (e - 3 * W)] = v;
  }
}

static inline bool ug_l3_ok(int W, const void *a, const void *b) {
  return W >= 4 && W <= 128 && (W & 3) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}

static int ug_l3_forward(const float *h2, int64_t M, int W, const float *w3, const float *b3, float *logits, hipStream_t st) {
  if (M <= 0) return 0;
  const int w4 = W >> 2;
  const int lpr = w4 <= 1 ? 1 : w4 <= 2 ? 2 : w4 <= 4 ? 4 : w4 <= 8 ? 8 : w4 <= 16 ? 16 : 32;
  const int64_t waves = (ug_launch_rows(M) + (UG_WAVE / lpr) - 1) / (UG_WAVE / lpr);
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 4096) blocks = 4096;
#define UG_L3F(L) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_l3_fwd<L>), dim3((unsigned)blocks), dim3(256), 0, st, h2, M, W, w3, b3, logits, ug_tl_devn.ptr)
  switch (lpr) {
    case 1: UG_L3F(1); break;
    case 2: UG_L3F(2); break;
    case 4: UG_L3F(4); break;
    case 8: UG_L3F(8); break;
    case 16: UG_L3F(16); break;
    default: UG_L3F(32); break;
  }
#undef UG_L3F
  UG_LAUNCH_CHECK();
  return 0;
}

// partial: UG_L3_MAX_BLOCKS * (3 W + 3) floats of scratch
static int ug_l3_backward(const float *g, const float *h2, int64_t M, int W, const float *w3, float *g_h2, float *dW, float *db,
                          float *partial, hipStream_t st) {
  if (M <= 0) {
    UG_HIP(hipMemsetAsync(dW, 0, sizeof(float) * 3 * (size_t)W, st));
    UG_HIP(hipMemsetAsync(db, 0, sizeof(float) * 3, st));
    return 0;
  }
  const int rpi = 256 / (W >> 2);
  const int64_t Ml = ug_launch_rows(M);                                  // (a device count in force: chunks and grid follow the hint)
  int64_t rows = 8 * (int64_t)rpi;                                       // >= 8 iterations per block
  if ((Ml + rows - 1) / rows > UG_L3_MAX_BLOCKS) rows = ((Ml + UG_L3_MAX_BLOCKS - 1) / UG_L3_MAX_BLOCKS + rpi - 1) / rpi * rpi;
  const int nb = (int)((Ml + rows - 1) / rows);
  float *pw = partial, *pb = partial + (size_t)UG_L3_MAX_BLOCKS * 3 * W;
  hipLaunchKernelGGL(k_l3_bwd, dim3((unsigned)nb), dim3(256), 0, st, g, h2, M, W, w3, g_h2, pw, pb, (int)rows, ug_tl_devn.ptr);
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((3 * W + 3 + 15) / 16), dim3(256), 0, st, pw, pb, nb, 3 * W, 3, dW, db);
  UG_LAUNCH_CHECK();
  return 0;
}

// ----------------------------------------------------------------------------------------------
// C ABI (declared in include/ugrid_hip.h)
// ----------------------------------------------------------------------------------------------
extern "C" int64_t ugrid_rgbnet_train_scratch_floats(int64_t M) {
  return 2 * M * 128 + (int64_t)UG_WG_MAX_SLABS * (128 * 128 + 128);
}

extern "C" int ugrid_rgbnet_train_forward(const float *feat, int64_t M, int32_t mlp_in, const float *w0, const float *b0,
                                          const float *w1, const float *b1, const float *w2, const float *b2, int32_t width,
                                          float *h1, float *h2, float *logits, ugrid_stream_t s) {
  if (width < 1 || width > 128 || mlp_in < 1 || mlp_in > 128) return (int)hipErrorNotSupported;
  const int W = width;
  int rc = ug_lin_launch(feat, M, mlp_in, mlp_in, w0, mlp_in, W, 0, b0, 1, nullptr, 0, h1, W, ST(s));
  if (rc) return rc;
  rc = ug_lin_launch(h1, M, W, W, w1, W, W, 0, b1, 1, nullptr, 0, h2, W, ST(s));
  if (rc) return rc;
  if (ug_l3_ok(W, h2, w2)) return ug_l3_forward(h2, M, W, w2, b2, logits, ST(s));
  return ug_lin_launch(h2, M, W, W, w2, W, 3, 0, b2, 0, nullptr, 0, logits, 3, ST(s));
}

extern "C" int ugrid_rgbnet_train_backward(const float *g_logits, const float *feat, const float *h1, const float *h2, int64_t M,
                                           int32_t mlp_in, int32_t n_feat_grad, const float *w0, const float *w1, const float *w2,
                                           int32_t width, float *g_feat, float *g_w0, float *g_b0, float *g_w1, float *g_b1,
                                           float *g_w2, float *g_b2, float *scratch, ugrid_stream_t s) {
  if (width < 1 || width > 128 || mlp_in < 1 || mlp_in > 128 || n_feat_grad < 0 || n_feat_grad > mlp_in) return (int)hipErrorNotSupported;
  const int W = width;
  float *g_h2 = scratch, *g_h1 = scratch + (size_t)M * 128;         // [M,W] each (room for W = 128)
  float *part = scratch + (size_t)2 * M * 128;                      // slab partials of the weight gradients
  int rc;
  if (ug_l3_ok(W, h2, w2) && ((uintptr_t)g_h2 & 15) == 0) {
    rc = ug_l3_backward(g_logits, h2, M, W, w2, g_h2, g_w2, g_b2, part, ST(s));                               // dW3, db3, dH2: one pass over h2
    if (rc) return rc;
  } else {
    rc = ug_wgrad_launch(g_logits, 3, 3, h2, W, W, M, g_w2, g_b2, part, ST(s));                              // dW3, db3
    if (rc) return rc;
    rc = ug_lin_launch(g_logits, M, 3, 3, w2, W, W, 1, nullptr, 0, h2, W, g_h2, W, ST(s));                    // dH2 = dL . W3, ReLU mask
    if (rc) return rc;
  }
  rc = ug_wgrad_launch(g_h2, W, W, h1, W, W, M, g_w1, g_b1, part, ST(s));                                     // dW2, db2
  if (rc) return rc;
  rc = ug_lin_launch(g_h2, M, W, W, w1, W, W, 1, nullptr, 0, h1, W, g_h1, W, ST(s));                          // dH1 = dH2 . W2, ReLU mask
  if (rc) return rc;
  rc = ug_wgrad_launch(g_h1, W, W, feat, mlp_in, mlp_in, M, g_w0, g_b0, part, ST(s));                         // dW1, db1
  if (rc) return rc;
  if (n_feat_grad > 0 && g_feat)                                                                            // d feat[:, :n] = dH1 . W1[:, :n]
    rc = ug_lin_launch(g_h1, M, W, W, w0, mlp_in, n_feat_grad, 1, nullptr, 0, nullptr, 0, g_feat, n_feat_grad, ST(s));
  return rc;
}
