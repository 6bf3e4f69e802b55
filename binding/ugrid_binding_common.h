// Shared helpers of the four pybind modules that bind libugrid_hip.so under the reference's extension-module names
// (INTEGRATION.md section B).  What a maintainer of the reference would commit in place of FourierGrid/cuda/*.cu: the
// same m.def() names, argument lists and return values as FourierGrid/cuda/render_utils.cpp:168-184,
// total_variation.cpp:21-24, ub360_utils.cpp:19-22, adam_upd.cpp:77-87 -- the device work goes to the C ABI of
// include/ugrid_hip.h on the tensor's device (device guard) and torch's CURRENT stream.
#pragma once
#include <torch/extension.h>
// torch-ROCm presents its HIP devices under the device type "cuda": the guard / stream classes that accept it are the
// "MasqueradingAsCUDA" ones (what torch's own hipify maps c10/cuda/CUDAGuard.h and CUDAStream.h to)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <vector>

#include "ugrid_hip.h"
#include "ugrid_hip_f64.h"      // the double instantiations (libugrid_hip_f64.so) of the ops the reference dispatches on the tensor type

// the reference's own input checks (render_utils.cpp:98-100), same wording
#define CHECK_CUDA(x) TORCH_CHECK(x.is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) CHECK_CUDA(x); CHECK_CONTIGUOUS(x)
#define CHECK_F32(x) TORCH_CHECK(x.scalar_type() == at::kFloat, #x " must be float32 (this op has no reference counterpart: fp32 only)")

// the two types of the reference's AT_DISPATCH_FLOATING_TYPES; one type per call (the reference reinterprets every array with the
// first tensor's)
#define CHECK_REAL(x) TORCH_CHECK(x.scalar_type() == at::kFloat || x.scalar_type() == at::kDouble, #x " must be float32 or float64")
#define CHECK_SAME(x, ref) TORCH_CHECK(x.scalar_type() == ref.scalar_type(), #x " and " #ref " must have one floating type")

static inline void ug_check(int err, const char *what) {
  TORCH_CHECK(err == 0, "libugrid_hip: ", what, " failed with hipError_t ", err);
}
static inline ugrid_stream_t ug_stream() { return (ugrid_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }
#define UG_GUARD(t) const c10::hip::HIPGuardMasqueradingAsCUDA ug_device_guard_((t).device())
static inline const float *fp(const torch::Tensor &t) { return t.data_ptr<float>(); }
static inline float *fpm(torch::Tensor &t) { return t.data_ptr<float>(); }
static inline bool is64(const torch::Tensor &t) { return t.scalar_type() == at::kDouble; }
static inline const double *dp(const torch::Tensor &t) { return t.data_ptr<double>(); }
static inline double *dpm(torch::Tensor &t) { return t.data_ptr<double>(); }
