// mfma_dep_hazard.hip -- minimal reproducer for the "dependent v_mfma_f32_32x32x16_{bf16,f16}" observation of DESIGN.md 4.1
// (VERDICT r2 "what's weak" 9: the product works around it with a pinned issue order + s_nop fences; this file is the
// stand-alone check a toolchain update can be tested with).
//
// Observation being tested: with ROCm 7.2's hipcc, a chain of MFMAs in which every instruction reads as SrcC the
// accumulator written by the MFMA issued IMMEDIATELY before it, with the B operands built by v_cvt_pk / v_fma_mix right
// before, occasionally produced run-to-run different results inside the shade kernel (one partial product lost, timing
// dependent).  The kernel below runs the same bf16x3-style k-steps in two issue orders
//   NAIVE  : tile-major -- the 6 products of a k-step on acc[0] back to back, then acc[1], ...  (dependent back-to-back MFMAs)
//   PINNED : round-robin over the 4 accumulator tiles (ug_mfma6x4's order), sched_barrier after each, s_nop fences around
//            the operand build
// on every CU at once, REPS times, and reports (a) run-to-run bitwise differences of each order against its own first
// run, (b) differences between the two orders (mathematically the same per-tile sequence of additions, so any difference
// is a hardware / compiler hazard, not rounding).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dep_hazard tools/microbench/mfma_dep_hazard.hip && /tmp/mfma_dep_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct split3 { bf16x8 h, m, l; };
__device__ __forceinline__ split3 split8(const float (&x)[8]) {
  split3 s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hh = (__bf16)x[i];
    const float r1 = x[i] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    s.h[i] = hh; s.m[i] = mm; s.l[i] = (__bf16)(r1 - (float)mm);
  }
  return s;
}

template <bool PINNED>
__global__ void __launch_bounds__(512, 2) k_chain(const float *__restrict__ wts, const float *__restrict__ xin, float *__restrict__ out, int ksteps) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  f32x16 acc[4];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[o][r] = 0.f;
  for (int ks = 0; ks < ksteps; ++ks) {
    float xv[8], wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xv[e] = xin[((size_t)ks * 64 + lane) * 8 + e] * (1.0f + 1e-3f * (float)(wave & 7));
      wv[e] = wts[((size_t)ks * 64 + lane) * 8 + e];
    }
    const split3 x = split8(xv), w = split8(wv);
    if (PINNED) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 15");
      __builtin_amdgcn_sched_barrier(0);
#define P(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, acc_, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
#pragma unroll
      for (int o = 0; o < 4; ++o) { P(acc[o], w.m, x.m); }
#pragma unroll
      for (int o = 0; o < 4; ++o) { P(acc[o], w.m, x.h); }
#pragma unroll
      for (int o = 0; o < 4; ++o) { P(acc[o], w.l, x.h); }
#pragma unroll
      for (int o = 0; o < 4; ++o) { P(acc[o], w.h, x.l); }
#pragma unroll
      for (int o = 0; o < 4; ++o) { P(acc[o], w.h, x.m); }
#pragma unroll
      for (int o = 0; o < 4; ++o) { P(acc[o], w.h, x.h); }
#undef P
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o) {     // the compiler is free to schedule these: dependent back-to-back MFMAs per tile
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, x.m, acc[o], 0, 0, 0);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, x.h, acc[o], 0, 0, 0);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, x.h, acc[o], 0, 0, 0);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x.l, acc[o], 0, 0, 0);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x.m, acc[o], 0, 0, 0);
        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x.h, acc[o], 0, 0, 0);
      }
    }
  }
  if (PINNED) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(wave * 64 + lane) * 64 + o * 16 + r] = acc[o][r];
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount, ksteps = 64, REPS = 40;
  const size_t n_in = (size_t)ksteps * 64 * 8, n_out = (size_t)blocks * 8 * 64 * 64;
  std::vector<float> hw(n_in), hx(n_in);
  unsigned s = 12345;
  for (size_t i = 0; i < n_in; ++i) {
    s = s * 1664525u + 1013904223u; hw[i] = ((s >> 8) / 16777216.0f - 0.5f) * 0.25f;
    s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) / 16777216.0f - 0.5f) * 4.0f;
  }
  float *dw, *dx, *dout;
  CHECK(hipMalloc(&dw, n_in * 4)); CHECK(hipMalloc(&dx, n_in * 4)); CHECK(hipMalloc(&dout, n_out * 4));
  CHECK(hipMemcpy(dw, hw.data(), n_in * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dx, hx.data(), n_in * 4, hipMemcpyHostToDevice));
  std::vector<float> first[2], cur(n_out);
  long long self_diff[2] = {0, 0}, runs_with_diff[2] = {0, 0};
  for (int rep = 0; rep < REPS; ++rep)
    for (int v = 0; v < 2; ++v) {
      if (v == 0) hipLaunchKernelGGL(k_chain<false>, dim3(blocks), dim3(512), 0, 0, dw, dx, dout, ksteps);
      else hipLaunchKernelGGL(k_chain<true>, dim3(blocks), dim3(512), 0, 0, dw, dx, dout, ksteps);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(cur.data(), dout, n_out * 4, hipMemcpyDeviceToHost));
      if (rep == 0) { first[v] = cur; continue; }
      long long d = 0;
      for (size_t i = 0; i < n_out; ++i) d += memcmp(&cur[i], &first[v][i], 4) != 0;
      self_diff[v] += d; runs_with_diff[v] += d != 0;
    }
  long long cross = 0;
  double max_abs = 0;
  for (size_t i = 0; i < n_out; ++i)
    if (memcmp(&first[0][i], &first[1][i], 4) != 0) { ++cross; double e = fabs((double)first[0][i] - first[1][i]); if (e > max_abs) max_abs = e; }
  printf("{\"device\": \"%s\", \"waves\": %d, \"ksteps\": %d, \"reps\": %d, \"values_per_run\": %zu, "
         "\"naive_order_runs_differing_from_first\": %lld, \"naive_order_values_differing\": %lld, "
         "\"pinned_order_runs_differing_from_first\": %lld, \"pinned_order_values_differing\": %lld, "
         "\"naive_vs_pinned_values_differing\": %lld, \"naive_vs_pinned_max_abs\": %.3e}\n",
         prop.gcnArchName, blocks * 8, ksteps, REPS, n_out, runs_with_diff[0], self_diff[0], runs_with_diff[1], self_diff[1], cross, max_abs);
  return 0;
}
