// Micro-benchmark (gfx950): can a SIMD execute one wave's MFMAs and another wave's VALU instructions at the same
// time?  One workgroup of 8 waves per CU (2 per SIMD: waves w and w+4 share SIMD w).  The lower four waves run
// role A, the upper four role B, each role being "mfma" (independent v_mfma_f32_32x32x16_f16, 4 accumulators),
// "valu" (independent v_fmac_f32, 8 chains) or "idle".  Prints the time of every combination; if
// t(mfma + valu) ~ max(t(mfma), t(valu)) the two pipes overlap across waves, if ~ sum they serialise.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_mfma_overlap valu_mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROLE>   // 0 idle, 1 mfma, 2 valu
__device__ __forceinline__ float run_role(int iters, float seed) {
  if (ROLE == 1) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {   // 32 MFMAs per iteration
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
      }
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
  }
  if (ROLE == 2) {
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
    const float m = 1.0000001f, k = 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {  // 256 FMAs per iteration (= 32 MFMAs x 8 passes of issue slots)
        x0 = fmaf(x0, m, k); x1 = fmaf(x1, m, k); x2 = fmaf(x2, m, k); x3 = fmaf(x3, m, k);
        x4 = fmaf(x4, m, k); x5 = fmaf(x5, m, k); x6 = fmaf(x6, m, k); x7 = fmaf(x7, m, k);
      }
    }
    return x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
  return seed;
}

template <int RA, int RB>
__global__ void __launch_bounds__(512, 2) k(int iters, float *out) {
  const int wave = threadIdx.x >> 6;
  float r = (wave < 4) ? run_role<RA>(iters, 1.0f + threadIdx.x) : run_role<RB>(iters, 2.0f + threadIdx.x);
  if (r == 123.456f) out[0] = r;   // keep the work alive
}

template <int RA, int RB>
static float timed(int iters, float *d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, iters / 10, d);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, iters, d);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *d; hipMalloc(&d, 4);
  const int it = 20000;
  const float mi = timed<1, 0>(it, d), vi = timed<2, 0>(it, d), mm = timed<1, 1>(it, d), vv = timed<2, 2>(it, d), mv = timed<1, 2>(it, d);
  printf("per SIMD, %d iterations of 32 MFMA (32x32x16 f16) / 256 v_fmac per wave\n", it);
  printf("mfma + idle   %.3f ms   (%.1f cycles per MFMA at 2.4 GHz)\n", mi, mi * 1e-3 * 2.4e9 / (it * 32.0));
  printf("valu + idle   %.3f ms   (%.2f cycles per FMA)\n", vi, vi * 1e-3 * 2.4e9 / (it * 256.0));
  printf("mfma + mfma   %.3f ms\n", mm);
  printf("valu + valu   %.3f ms\n", vv);
  printf("mfma + valu   %.3f ms   -> overlap = %.2f   (1 = perfect: max of the two; 0 = serialised: their sum)\n", mv,
         (mi + vi - mv) / (mi + vi - (mi > vi ? mi : vi)));
  return 0;
}
