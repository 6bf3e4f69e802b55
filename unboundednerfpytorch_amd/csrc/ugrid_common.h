// Shared helpers for the gfx950 kernels of libugrid_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugrid_hip.h"

#define UG_WAVE 64

#define UG_LAUNCH_CHECK()                         \
  do {                                            \
    hipError_t _e = hipGetLastError();            \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

#define UG_HIP(expr)                              \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

static inline unsigned ug_blocks(int64_t n, int threads) {
  return (unsigned)((n + threads - 1) / threads);
}

__device__ __forceinline__ int ug_lane() { return (int)(threadIdx.x & (UG_WAVE - 1)); }

// Uniform broadcast of lane `k` (k wave-uniform) of a 32-bit / 64-bit value.
__device__ __forceinline__ float ug_readlane_f(float v, int k) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}
__device__ __forceinline__ double ug_readlane_d(double v, int k) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), k);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), k);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
