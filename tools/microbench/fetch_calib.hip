// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE (and the TCC_EA0_RDREQ_{32B,64B,128B} request counters it is derived from)
// report for the access SHAPES of the render kernels?  VERDICT r3 item 3(b): the guide calibrates FETCH_SIZE x 2 only for wide
// coalesced 16 B/lane streams and calls every other width uncalibrated; the march gathers 32-byte records, the shade 384-byte
// records in 64-byte quad pieces.
//
// Four kernels read a KNOWN number of bytes, each byte once, from an 8 GiB array (32 x the 256 MiB Infinity Cache, so nothing
// is served on-die twice):
//   k_linear   every lane 16 B, consecutive lanes consecutive addresses (the guide's calibrated case)
//   k_rec32    every lane one 32-byte record (two dwordx4) at a scattered, 32-byte aligned place   (march: density bricks)
//   k_rec64    every lane quad one 64-byte piece (4 x 16 B) at a scattered, 64-byte aligned place
//   k_rec384   every lane quad one 384-byte record (6 x 64 B) at a scattered, 384-byte aligned place (shade: k0 quad bricks)
// "scattered": record index = (i * ODD) mod N with N a power of two -- a permutation, every record exactly once.
// Run under   rocprofv3 --kernel-trace --pmc FETCH_SIZE                                              -- ./fetch_calib
//       and   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -- ./fetch_calib
// the program prints the requested bytes per kernel as JSON; tools/microbench/fetch_calib_report.py joins the two.
//   hipcc --offload-arch=gfx950 -O3 -o build/ab/fetch_calib tools/microbench/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_linear(const f4 *__restrict__ a, uint64_t n16, float *sink) {
  f4 s = {0, 0, 0, 0};
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) s += a[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) *sink = s.x;
}
__global__ void __launch_bounds__(256) k_rec32(const f4 *__restrict__ a, uint64_t nrec, uint64_t odd, float *sink) {
  f4 s = {0, 0, 0, 0};
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = (i * odd) & (nrec - 1);
    s += a[2 * r] + a[2 * r + 1];
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) *sink = s.x;
}
// lane quad q reads piece `PIECES` x 64 B of record r: lane g of the quad its own 16 bytes of each piece
template <int PIECES>
__global__ void __launch_bounds__(256) k_recq(const f4 *__restrict__ a, uint64_t nrec, uint64_t odd, float *sink) {
  f4 s = {0, 0, 0, 0};
  const uint64_t quads = ((uint64_t)gridDim.x * blockDim.x) >> 2;
  const int g = threadIdx.x & 3;
  for (uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2; q < nrec; q += quads) {
    const uint64_t r = (q * odd) & (nrec - 1);
#pragma unroll
    for (int p = 0; p < PIECES; ++p) s += a[(r * PIECES + p) * 4 + g];
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) *sink = s.x;
}

int main(int argc, char **argv) {
  const uint64_t GiB = 1ull << 30;
  const uint64_t bytes = 8 * GiB;
  f4 *a; float *sink;
  CK(hipMalloc(&a, bytes + 4096)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, bytes));
  CK(hipDeviceSynchronize());
  const int grid = 256 * 8;
  const uint64_t odd = 0x9E3779B97F4A7C15ull | 1ull;
  // each kernel touches 4 GiB of the array exactly once (k_rec384: 2^23 records of 384 B = 3 GiB)
  const uint64_t n16 = 4 * GiB / 16, n32 = 4 * GiB / 32, n64 = 4 * GiB / 64, n384 = 1ull << 23;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_linear, dim3(grid), dim3(256), 0, 0, a, n16, sink);
    hipLaunchKernelGGL(k_rec32, dim3(grid), dim3(256), 0, 0, a + (4 * GiB / 16), n32, odd, sink);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_recq<1>), dim3(grid), dim3(256), 0, 0, a, n64, odd, sink);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_recq<6>), dim3(grid), dim3(256), 0, 0, a + (4 * GiB / 16), n384, odd, sink);
    CK(hipDeviceSynchronize());
  }
  printf("{\"requested_bytes\": {\"k_linear\": %llu, \"k_rec32\": %llu, \"k_recq<1>\": %llu, \"k_recq<6>\": %llu}}\n",
         (unsigned long long)(n16 * 16), (unsigned long long)(n32 * 32), (unsigned long long)(n64 * 64), (unsigned long long)(n384 * 384));
  return 0;
}
