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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's copy of the function: remember it per
// (call site, device), not per process -- a second GPU in the process would otherwise launch with the 64 KB default and
// fail.  `done` is the call site's own `static unsigned long long` bit set (devices 0..63); racing threads (forward on the
// main thread, backward on the autograd thread) at worst set the attribute twice, which is harmless.
#define UG_SET_DYN_LDS(func, bytes)                                                                           \
  do {                                                                                                        \
    static unsigned long long done_ = 0ull;                                                                   \
    int dev_ = 0;                                                                                             \
    UG_HIP(hipGetDevice(&dev_));                                                                              \
    const unsigned long long bit_ = 1ull << (dev_ & 63);                                                      \
    if (dev_ > 63 || !(__atomic_load_n(&done_, __ATOMIC_RELAXED) & bit_)) {                                   \
      UG_HIP(hipFuncSetAttribute((const void *)(func), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
      __atomic_fetch_or(&done_, bit_, __ATOMIC_RELAXED);                                                      \
    }                                                                                                         \
  } while (0)

// ---- device-resident row counts (the sync-free training step, ugrid_step.hip) -------------------------------------------------
// A training step's per-sample arrays have M1 / M2 rows, numbers that exist only on the device until somebody reads them.  While
// a ug_devn_scope is alive on the calling thread, the launchers of the per-sample training kernels (grid query fwd / bwd, rgbnet
// features, k_lin*, k_wgrad*, k_l3_*, the loss kernels) hand `ptr` to their kernels, which then process min(n, *ptr) rows in
// grid-stride loops: the host's `n` is only the CAPACITY of the arrays, and `hint` (> 0: e.g. the previous step's count) sizes the
// grids.  Without a scope the kernels get a null pointer and behave exactly as before.
struct ug_devn { const int64_t *ptr; int64_t hint; int64_t cap; };      // cap (> 0): rows the stage-2 arrays hold (compaction drops what exceeds it)
extern thread_local ug_devn ug_tl_devn;               // defined in ugrid_step.hip
struct ug_devn_scope {
  ug_devn saved;
  ug_devn_scope(const int64_t *p, int64_t hint, int64_t cap = 0) : saved(ug_tl_devn) { ug_tl_devn.ptr = p; ug_tl_devn.hint = hint; ug_tl_devn.cap = cap; }
  ~ug_devn_scope() { ug_tl_devn = saved; }
};
// rows the grid of a launch is sized for: the capacity, or the hint where a device count will stop the kernel anyway
static inline int64_t ug_launch_rows(int64_t n) {
  return (ug_tl_devn.ptr && ug_tl_devn.hint > 0 && ug_tl_devn.hint < n) ? ug_tl_devn.hint : n;
}
#define UG_DEVN_CLAMP(n, n_dev)                 \
  do {                                          \
    if (n_dev) {                                \
      const int64_t nd_ = *(n_dev);             \
      if (nd_ < (n)) (n) = nd_ < 0 ? 0 : nd_;   \
    }                                           \
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
