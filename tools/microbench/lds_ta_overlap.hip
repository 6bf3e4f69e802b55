// Micro-benchmark (gfx950): do vector-memory gathers (texture addresser / L1 return path) and LDS reads overlap inside
// a CU, or do they share a return path?  One workgroup of 16 waves per CU; waves 0-7 run role A, waves 8-15 role B:
//   "vmem": global_load_dwordx4, every quad of lanes reads 64 contiguous bytes of a random 384-byte record of an
//           L2-resident table (the access pattern of the quad k0 gather), 6 loads in flight
//   "lds" : ds_read_b128 of a per-lane 16-byte slot (conflict-free, the A-operand pattern of the rgbnet), 6 in flight
//   "idle"
// If t(vmem + lds) ~ max(t(vmem), t(lds)) the two paths are independent; if ~ sum, they share a resource.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_ta_overlap lds_ta_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, const float *__restrict__ table, unsigned nrec, float *lds, unsigned seed) {
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (ROLE == 1) {
    unsigned s = seed * 2654435761u + (lane >> 2) * 40503u;
    for (int it = 0; it < iters; ++it) {
      s = s * 1664525u + 1013904223u;
      const unsigned rec = (s >> 8) % nrec;
      const unsigned off = rec * 384u + (lane & 3) * 16u;
      f4 v0, v1, v2, v3, v4, v5;
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:0" : "=v"(v0) : "v"(off), "s"(table) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(v1) : "v"(off), "s"(table) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "=v"(v2) : "v"(off), "s"(table) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:192" : "=v"(v3) : "v"(off), "s"(table) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:256" : "=v"(v4) : "v"(off), "s"(table) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:320" : "=v"(v5) : "v"(off), "s"(table) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5));
      acc += v0.x + v1.y + v2.z + v3.w + v4.x + v5.y;
    }
  }
  if (ROLE == 2) {
    const f4 *p = (const f4 *)lds + lane;
    for (int it = 0; it < iters; ++it) {
      const f4 *q = p + ((it & 7) * 6) * 64;
      const f4 v0 = q[0], v1 = q[64], v2 = q[128], v3 = q[192], v4 = q[256], v5 = q[320];
      asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5));
      acc += v0.x + v1.y + v2.z + v3.w + v4.x + v5.y;
    }
  }
  return acc + seed;
}

template <int RA, int RB>
__global__ void __launch_bounds__(1024, 1) k(int iters, const float *table, unsigned nrec, float *out) {
  extern __shared__ float lds[];   // 64 KB
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const unsigned seed = blockIdx.x * 16 + wave;
  const float r = (wave < 8) ? run_role<RA>(iters, table, nrec, lds, seed) : run_role<RB>(iters, table, nrec, lds, seed);
  if (r == 123.456f) out[0] = r;
}

template <int RA, int RB>
static float timed(int iters, const float *table, unsigned nrec, float *d) {
  hipFuncSetAttribute((const void *)k<RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(1024), 65536, 0, iters / 10, table, nrec, d);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(1024), 65536, 0, iters, table, nrec, d);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const unsigned nrec = 8192;   // 3 MB table: stays in every XCD's 4 MB L2
  float *table, *d;
  hipMalloc(&table, (size_t)nrec * 384);
  hipMemset(table, 0, (size_t)nrec * 384);
  hipMalloc(&d, 4);
  const int it = 20000;
  const float vi = timed<1, 0>(it, table, nrec, d), li = timed<2, 0>(it, table, nrec, d), vv = timed<1, 1>(it, table, nrec, d),
              ll = timed<2, 2>(it, table, nrec, d), vl = timed<1, 2>(it, table, nrec, d);
  const double bytes = 8.0 * 256 * it * 6 * 1024;   // 8 waves x 256 CUs x iters x 6 x 1 KiB per role
  printf("8 waves per CU per role, %d iterations of 6 x (64 lanes x 16 B)\n", it);
  printf("vmem + idle   %.3f ms   %.1f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", vi, bytes / vi / 1e9, bytes / (vi * 1e-3) / 256 / 2.4e9);
  printf("lds  + idle   %.3f ms   %.1f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", li, bytes / li / 1e9, bytes / (li * 1e-3) / 256 / 2.4e9);
  printf("vmem + vmem   %.3f ms   %.1f TB/s\n", vv, 2 * bytes / vv / 1e9);
  printf("lds  + lds    %.3f ms   %.1f TB/s\n", ll, 2 * bytes / ll / 1e9);
  printf("vmem + lds    %.3f ms   -> overlap = %.2f  (1 = independent paths: max of the two; 0 = serialised: their sum)\n", vl,
         (vi + li - vl) / (vi + li - (vi > li ? vi : li)));
  return 0;
}
