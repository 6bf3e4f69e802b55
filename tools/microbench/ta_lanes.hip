// ta_lanes.hip -- what does a gather cost on the CU's vector-memory path (TA / TCP), and what would the alternatives cost?
//
// VERDICT r4 item 2(a).  Both frame kernels push their bytes through the per-CU vector-memory path (TA busy 0.90 / 0.80) while
// the 64 lanes of a march load hit 1-6 distinct 32-byte cell records.  Three questions decide whether de-duplicating those
// fetches can pay:
//   1. does the TA's time per `global_load_dwordx4` wave instruction scale with the ACTIVE lanes (or active quads), or is it a
//      flat 16 clocks per instruction?                                   -> section "ta": lane masks x address shapes
//   2. do lanes that read the SAME record cost less than lanes that read different ones?  -> shapes same / quad / cells4 / own32
//   3. what do the other delivery routes cost per wave instruction: `ds_read_b128` from LDS (incl. broadcast and the bank
//      conflicts of scattered 32-byte records), `ds_bpermute_b32`, scalar `s_load_dwordx8`, and a whole "LDS-staged" sample
//      (a few `global_load_lds_dwordx4` that land a cell neighbourhood in LDS + 14 `ds_read_b128`) against the march's 14
//      `global_load_dwordx4`?                                            -> sections "lds", "smem", "sample"
// Every CU runs 8 waves (2 workgroups x 4) on an L1-resident window, nothing but the unit
// under test limits the rate; clocks = s_memtime ticks (= shader cycles) of the longest wave; "clk_per_wave_instr" = those
// cycles / (wave instructions issued on one CU).  Prints one JSON object.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/ab/ta_lanes tools/microbench/ta_lanes.hip && build/ab/ta_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u8v __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { OWN32 = 0, SAME = 1, QUAD = 2, CELLS4 = 3, LINEAR = 4, CELLS6Z = 5 };
static const char *SHAPE_NAME[] = {"own32", "same", "quad", "cells4", "linear", "cells6_zorder"};

// byte offset (inside an 8 KiB window) of the 16-byte piece lane `lane` reads for load k of iteration it
template <int SHAPE>
__device__ __forceinline__ unsigned piece_offset(int lane, int wave, int it, int k) {
  const unsigned rec_i = (unsigned)(it * 4 + (k >> 1)), half = (unsigned)(k & 1) * 16u;
  unsigned key;
  if (SHAPE == OWN32) key = lane * 97u;                       // every lane its own 32-byte record (the march's shape today)
  else if (SHAPE == SAME) key = 0u;                           // all lanes the same record
  else if (SHAPE == QUAD) key = (lane >> 2) * 97u;            // the four lanes of a quad share a record
  else if (SHAPE == CELLS4) key = (lane >> 4) * 97u;          // four records per wave instruction, 16-lane runs
  else if (SHAPE == CELLS6Z) {                                // ~6 records, assigned by pixel position in the 8 x 8 Z-order block
    const int x = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), y = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    key = ((x >= 3) + 2 * (y >= 5) + 4 * ((x + y) >= 11)) * 97u;
  } else return (unsigned)(lane * 16 + (((it * 8 + k) * 4 + wave) & 7) * 1024);      // LINEAR: one coalesced KiB
  const unsigned q = key + rec_i * 31u + wave * 13u;
  return ((q * 2654435761u) >> 24) * 32u + half;              // 256 x 32 B = 8 KiB
}

// ---------------------------------------------------------------------------------------------------------------
// 1 + 2: global_load_dwordx4 under a lane mask
// ---------------------------------------------------------------------------------------------------------------
template <int SHAPE>
__global__ void __launch_bounds__(256) k_ta(const float *__restrict__ buf, int iters, unsigned long long mask,
                                            float *__restrict__ sink, unsigned long long *__restrict__ cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char *win = (const char *)buf + (size_t)blockIdx.x * 8192;
  float acc = 0.f;
  const bool active = (mask >> lane) & 1ull;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (active) {
    for (int it = 0; it < iters; ++it) {
      f4 v[8];
      unsigned off[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) off[k] = piece_offset<SHAPE>(lane, wave, it, k);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[k]) : "v"(off[k]), "s"(win) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k].x;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
  if (lane == 0) atomicMax(cycles, t1 - t0);
}

// ---------------------------------------------------------------------------------------------------------------
// 3a: ds_read_b128 (MODE 0) / ds_bpermute_b32 (MODE 1) with the same address shapes
// ---------------------------------------------------------------------------------------------------------------
template <int SHAPE, int MODE>
__global__ void __launch_bounds__(256) k_lds(const float *__restrict__ buf, int iters, float *__restrict__ sink,
                                             unsigned long long *__restrict__ cycles) {
  __shared__ f4 sm[512];                                  // 8 KiB per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sm[threadIdx.x] = ((const f4 *)buf)[threadIdx.x];
  sm[threadIdx.x + 256] = ((const f4 *)buf)[threadIdx.x + 256];
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(void *)sm;     // 32-bit LDS byte address
  float acc = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      f4 v[8];
      unsigned off[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) off[k] = base + piece_offset<SHAPE>(lane, wave, it, k);
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(off[k]) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k].x;
    } else {
      float v[8];
      const unsigned src = ((piece_offset<SHAPE>(lane, wave, it, 0) >> 5) & 63u) << 2;     // source lane * 4
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[k]) : "v"(src), "v"(acc + (float)k) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
  if (lane == 0) atomicMax(cycles, t1 - t0);
}

// ---------------------------------------------------------------------------------------------------------------
// 3b: scalar path -- s_load_dwordx8 of a wave-uniform 32-byte record (what a "cell-uniform wave" could use)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_smem(const float *__restrict__ buf, int iters, float *__restrict__ sink,
                                              unsigned long long *__restrict__ cycles) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char *win = (const char *)buf + (size_t)blockIdx.x * 8192;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    u8v v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned q = (unsigned)(it * 4 + k) * 31u + wave * 13u;
      const unsigned off = ((q * 2654435761u) >> 24) * 32u;
      asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v[k]) : "s"(win), "s"(off) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v[0]), "+s"(v[1]), "+s"(v[2]), "+s"(v[3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += v[k][0] + v[k][7];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 0x12345678u) sink[threadIdx.x] = (float)acc;
  if (lane == 0) atomicMax(cycles, t1 - t0);
}

// ---------------------------------------------------------------------------------------------------------------
// 3c: one whole march "sample" (7 levels, 14 pieces of 16 B per lane) by three delivery routes, no arithmetic:
//   MODE 0  today: 14 global_load_dwordx4, the lanes' records drawn from NC distinct cells per level (cells6_zorder)
//   MODE 1  staged 3x3x3: per level one global_load_lds_dwordx4 lands 1 KiB (a 27-cell neighbourhood = 864 B) in the wave's LDS
//           slot, then 14 ds_read_b128 from it (same cell assignment); double-buffered over samples
//   MODE 2  staged 2x2x2: 2 global_load_lds_dwordx4 per sample land all 7 levels' 8-cell neighbourhoods (7 x 256 B), then 14 reads
// 8 waves per CU (2 workgroups x 4, both resident: 56 KiB of LDS each); LDS: 2 buffers x 7 KiB per wave.
// ---------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) k_sample(const float *__restrict__ buf, int iters, float *__restrict__ sink,
                                                unsigned long long *__restrict__ cycles) {
  extern __shared__ f4 smx[];                              // 4 waves x 2 buffers x 7 KiB = 56 KiB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char *win = (const char *)buf + (size_t)blockIdx.x * 8192;
  // the lane's cell among the neighbourhood (constant over the run, like a pixel's place in the 8 x 8 block)
  const int x = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), y = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
  const unsigned cell = (x >= 3) + 2 * (y >= 5) + 4 * ((x + y) >= 11);      // 0..7: 6 of them occur
  float acc = 0.f;
  const unsigned my_lds = (unsigned)(size_t)(void *)smx + (unsigned)wave * 2u * 7168u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      f4 v[14];
#pragma unroll
      for (int l = 0; l < 7; ++l) {
        const unsigned q = cell * 97u + (unsigned)(it * 7 + l) * 31u + wave * 13u;
        const unsigned off = ((q * 2654435761u) >> 24) * 32u;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[2 * l]) : "v"(off), "s"(win) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(v[2 * l + 1]) : "v"(off), "s"(win) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                   "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]));
#pragma unroll
      for (int k = 0; k < 14; ++k) acc += v[k].x;
    }
  } else {
    constexpr int NST = (MODE == 1) ? 7 : 2;               // LDS-DMA pieces (1 KiB each) per sample
    auto stage = [&](int it, unsigned dst) {
#pragma unroll
      for (int s = 0; s < NST; ++s) {
        const unsigned q = (unsigned)(it * NST + s) * 31u + wave * 13u;
        const unsigned off = (((q * 2654435761u) >> 29) * 1024u) + (unsigned)lane * 16u;    // a contiguous KiB of the window
        const unsigned d = dst + (unsigned)s * 1024u;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(off), "s"(d), "s"(win) : "memory");
      }
    };
    stage(0, my_lds);
    for (int it = 0; it < iters; ++it) {
      const unsigned cur = my_lds + (unsigned)(it & 1) * 7168u, nxt = my_lds + (unsigned)((it + 1) & 1) * 7168u;
      stage(it + 1, nxt);
      if (MODE == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      f4 v[14];
#pragma unroll
      for (int l = 0; l < 7; ++l) {
        // the lane's record inside level l's staged neighbourhood: MODE 1: 27 cells per KiB slot; MODE 2: 8 cells per 256 B
        const unsigned a = cur + (MODE == 1 ? (unsigned)l * 1024u + ((cell * 3u + (unsigned)l) % 27u) * 32u
                                            : (unsigned)l * 256u + cell * 32u);
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[2 * l]) : "v"(a) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(v[2 * l + 1]) : "v"(a) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                   "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]));
#pragma unroll
      for (int k = 0; k < 14; ++k) acc += v[k].x;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 12345.678f) sink[threadIdx.x] = acc;
  if (lane == 0) atomicMax(cycles, t1 - t0);
}

// ---------------------------------------------------------------------------------------------------------------
struct Ctx { float *buf, *sink; unsigned long long *d_cyc; int cus, blocks; };

template <typename F>
static void timed(const Ctx &c, F launch, double *cyc_out, double *ms_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double best_cyc = 0, best_ms = 0;
  for (int rep = 0; rep < 3; ++rep) {          // keep the last: window warm, clock settled
    CHECK(hipMemset(c.d_cyc, 0, 8));
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cyc = 0;
    CHECK(hipMemcpy(&cyc, c.d_cyc, 8, hipMemcpyDeviceToHost));
    best_cyc = (double)cyc; best_ms = ms;
  }
  *cyc_out = best_cyc; *ms_out = best_ms;
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

template <int SHAPE>
static double ta_case(const Ctx &c, unsigned long long mask, int iters) {
  double cyc, ms;
  timed(c, [&] { hipLaunchKernelGGL(k_ta<SHAPE>, dim3(c.blocks), dim3(256), 0, 0, c.buf, iters, mask, c.sink, c.d_cyc); }, &cyc, &ms);
  return cyc / ((double)iters * 8.0 * 8.0);    // 8 waves per CU x 8 loads per iteration
}

static double ta_dispatch(const Ctx &c, int shape, unsigned long long mask, int iters) {
  switch (shape) {
    case OWN32: return ta_case<OWN32>(c, mask, iters);
    case SAME: return ta_case<SAME>(c, mask, iters);
    case QUAD: return ta_case<QUAD>(c, mask, iters);
    case CELLS4: return ta_case<CELLS4>(c, mask, iters);
    case CELLS6Z: return ta_case<CELLS6Z>(c, mask, iters);
    default: return ta_case<LINEAR>(c, mask, iters);
  }
}

template <int SHAPE, int MODE>
static double lds_case(const Ctx &c, int iters) {
  double cyc, ms;
  timed(c, [&] { hipLaunchKernelGGL((k_lds<SHAPE, MODE>), dim3(c.blocks), dim3(256), 0, 0, c.buf, iters, c.sink, c.d_cyc); }, &cyc, &ms);
  return cyc / ((double)iters * 8.0 * 8.0);
}

template <int MODE>
static double sample_case(const Ctx &c, int iters, double *ghz) {
  double cyc, ms;
  CHECK(hipFuncSetAttribute((const void *)k_sample<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 7168));
  // the same 56 KiB for every mode, so that all three run with the same residency: 2 workgroups = 8 waves per CU
  timed(c, [&] { hipLaunchKernelGGL(k_sample<MODE>, dim3(c.blocks), dim3(256), 4 * 2 * 7168, 0, c.buf, iters, c.sink, c.d_cyc); },
        &cyc, &ms);
  *ghz = cyc / (ms * 1e-3) / 1e9;              // (= the shader clock only if every workgroup was resident at once)
  return cyc / ((double)iters * 8.0);          // clocks per wave-sample per CU (8 waves per CU)
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  Ctx c;
  c.cus = prop.multiProcessorCount;
  c.blocks = c.cus * 2;
  CHECK(hipMalloc(&c.buf, (size_t)c.blocks * 8192 + 4096));
  CHECK(hipMemset(c.buf, 0, (size_t)c.blocks * 8192 + 4096));
  CHECK(hipMalloc(&c.sink, 4096)); CHECK(hipMalloc(&c.d_cyc, 8));
  const int iters = 2048;

  struct M { const char *name; unsigned long long mask; int lanes; };
  std::vector<M> masks = {
    {"all64", ~0ull, 64}, {"first32", 0xFFFFFFFFull, 32}, {"first16", 0xFFFFull, 16}, {"first8", 0xFFull, 8},
    {"first4_one_quad", 0xFull, 4}, {"first1", 0x1ull, 1},
    {"every_2nd_quad_32", 0x0F0F0F0F0F0F0F0Full, 32}, {"one_per_quad_16", 0x1111111111111111ull, 16},
    {"two_per_quad_32", 0x3333333333333333ull, 32}, {"one_per_8_lanes_8", 0x0101010101010101ull, 8},
    {"one_per_16_lanes_4", 0x0001000100010001ull, 4},
  };
  printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_cu\": 8, \"loads_in_flight_per_wave\": 8, \"unit\": \"shader clocks per wave instruction per CU\",\n", prop.gcnArchName, c.cus);
  printf(" \"ta_global_load_dwordx4\": {\n");
  const int shapes[] = {OWN32, QUAD, CELLS6Z, CELLS4, SAME, LINEAR};
  for (size_t s = 0; s < sizeof(shapes) / sizeof(int); ++s) {
    printf("  \"%s\": {", SHAPE_NAME[shapes[s]]);
    for (size_t m = 0; m < masks.size(); ++m)
      printf("%s\"%s\": %.2f", m ? ", " : "", masks[m].name, ta_dispatch(c, shapes[s], masks[m].mask, iters));
    printf("}%s\n", s + 1 < sizeof(shapes) / sizeof(int) ? "," : "");
  }
  printf(" },\n \"lds_ds_read_b128\": {\"own32\": %.2f, \"quad\": %.2f, \"cells6_zorder\": %.2f, \"cells4\": %.2f, \"same\": %.2f, \"linear\": %.2f},\n",
         lds_case<OWN32, 0>(c, iters), lds_case<QUAD, 0>(c, iters), lds_case<CELLS6Z, 0>(c, iters), lds_case<CELLS4, 0>(c, iters),
         lds_case<SAME, 0>(c, iters), lds_case<LINEAR, 0>(c, iters));
  printf(" \"lds_ds_bpermute_b32\": {\"own32\": %.2f, \"quad\": %.2f, \"same\": %.2f},\n",
         lds_case<OWN32, 1>(c, iters), lds_case<QUAD, 1>(c, iters), lds_case<SAME, 1>(c, iters));
  {
    double cyc, ms;
    timed(c, [&] { hipLaunchKernelGGL(k_smem, dim3(c.blocks), dim3(256), 0, 0, c.buf, iters, c.sink, c.d_cyc); }, &cyc, &ms);
    printf(" \"smem_s_load_dwordx8\": {\"clk_per_instr_per_cu\": %.2f, \"note\": \"8 waves per CU, 4 in flight per wave, scalar-cache resident\"},\n",
           cyc / ((double)iters * 4.0 * 8.0));
  }
  double g0, g1, g2;
  const double s0 = sample_case<0>(c, 1024, &g0), s1 = sample_case<1>(c, 1024, &g1), s2 = sample_case<2>(c, 1024, &g2);
  printf(" \"sample_7_levels_14_pieces\": {\"waves_per_cu\": 8, \"unit\": \"shader clocks per wave-sample per CU\", "
         "\"today_14_global_load_dwordx4\": %.1f, \"staged_3x3x3_7_lds_dma_plus_14_ds_read\": %.1f, \"staged_2x2x2_2_lds_dma_plus_14_ds_read\": %.1f, "
         "\"clock_GHz\": [%.3f, %.3f, %.3f]}\n}\n", s0, s1, s2, g0, g1, g2);
  return 0;
}
