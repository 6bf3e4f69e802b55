// l1_dwordx4.hip -- calibration of the vector-L1 (texture path) peak that bench.py's roofline block prices the gather
// kernels against (VERDICT r2 "what's weak" 4b: the nominal 64 B/clk/CU is not a figure of MI355X_MICROARCH.md).
//
// Every CU streams global_load_dwordx4 from a window that stays resident in its 32 KiB L1 (8 KiB per workgroup, 2
// workgroups per CU), so nothing but the TA / L1 data path limits the rate.  Two access shapes:
//   linear : lane i reads 16 B at 16 i + 1 KiB k                (one fully coalesced KiB per wave instruction)
//   quad64 : every lane quad reads 64 contiguous, 64-byte aligned bytes at a pseudo-random place of the window
//            (the shade kernel's k0 gather: 16 distinct 64-byte pieces per wave instruction)
//   pair32 : every lane reads two dwordx4 = its own 32-byte record at a pseudo-random place (the march kernel's bricks)
// Prints one JSON object: bytes per clock per CU at the clock the chip actually ran (s_memtime ticks = shader cycles).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l1_dwordx4 tools/microbench/l1_dwordx4.hip && /tmp/l1_dwordx4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_l1(const float *__restrict__ buf, int iters, float *__restrict__ sink,
                                            unsigned long long *__restrict__ cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char *win = (const char *)buf + (size_t)blockIdx.x * 8192;       // this workgroup's L1-resident window
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned seed = lane * 2654435761u + wave * 40503u + 12345u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    f4 v[8];
    unsigned off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) {
        off[k] = ((unsigned)(lane * 16 + (((it * 8 + k) * 4 + wave) & 7) * 1024));
      } else if (MODE == 1) {
        const unsigned q = (lane >> 2) * 97u + (it * 8 + k) * 31u + wave * 13u;
        off[k] = ((q * 2654435761u) >> 25) * 64u + (lane & 3) * 16u;      // 128 x 64 B = 8 KiB
      } else {
        const unsigned q = lane * 97u + (it * 4 + (k >> 1)) * 31u + wave * 13u;
        off[k] = ((q * 2654435761u) >> 24) * 32u + (k & 1) * 16u;         // 256 x 32 B = 8 KiB
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[k]) : "v"(off[k]), "s"(win) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[threadIdx.x] = acc.x;   // keep the loads alive
  if (lane == 0) atomicMax(cycles, t1 - t0);
  (void)seed;
}

template <int MODE>
static double run(const float *buf, int blocks, int iters, float *sink, unsigned long long *d_cyc, double *cyc_out, double *ms_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_l1<MODE>, dim3(blocks), dim3(256), 0, 0, buf, 16, sink, d_cyc);     // warm-up: fills the L1 windows
  CHECK(hipMemset(d_cyc, 0, 8));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_l1<MODE>, dim3(blocks), dim3(256), 0, 0, buf, iters, sink, d_cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc = 0;
  CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
  *cyc_out = (double)cyc; *ms_out = ms;
  return (double)blocks * 256.0 * iters * 8.0 * 16.0;      // bytes requested
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 2, iters = 4096;
  float *buf, *sink; unsigned long long *d_cyc;
  CHECK(hipMalloc(&buf, (size_t)blocks * 8192));
  CHECK(hipMemset(buf, 0, (size_t)blocks * 8192));
  CHECK(hipMalloc(&sink, 1024)); CHECK(hipMalloc(&d_cyc, 8));
  const char *names[3] = {"linear", "quad64", "pair32"};
  double bpc[3], gbs[3], ghz[3];
  for (int m = 0; m < 3; ++m) {
    double cyc, ms, bytes = 0;
    for (int rep = 0; rep < 3; ++rep) {    // keep the last (clock settled)
      if (m == 0) bytes = run<0>(buf, blocks, iters, sink, d_cyc, &cyc, &ms);
      if (m == 1) bytes = run<1>(buf, blocks, iters, sink, d_cyc, &cyc, &ms);
      if (m == 2) bytes = run<2>(buf, blocks, iters, sink, d_cyc, &cyc, &ms);
    }
    bpc[m] = bytes / cyc / cus;            // longest wave's cycles ~ the kernel's cycles on every CU
    gbs[m] = bytes / (ms * 1e-3) / 1e9;
    ghz[m] = cyc / (ms * 1e-3) / 1e9;
  }
  printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_cu\": 8, \"window_bytes_per_cu\": 16384, \"loads_in_flight_per_wave\": 8", prop.gcnArchName, cus);
  for (int m = 0; m < 3; ++m)
    printf(", \"%s_B_per_clk_per_CU\": %.2f, \"%s_GBps\": %.0f, \"%s_clock_GHz\": %.3f", names[m], bpc[m], names[m], gbs[m], names[m], ghz[m]);
  printf(", \"nominal_B_per_clk_per_CU\": 64}\n");
  return 0;
}
