// total_variation_cuda for MI355X (FourierGrid/cuda/total_variation.cpp:21-24).
#include "ugrid_binding_common.h"

// grad += TV gradient of param, in place; sizes from param.size(2..4) (total_variation_kernel.cu:39-41)
void total_variation_add_grad(torch::Tensor param, torch::Tensor grad, float wx, float wy, float wz, bool dense_mode) {
  CHECK_INPUT(param); CHECK_INPUT(grad); CHECK_REAL(param); CHECK_SAME(grad, param);
  TORCH_CHECK(param.dim() == 5 && param.sizes() == grad.sizes(), "param/grad must be 5-D tensors of equal shape");
  UG_GUARD(param);
  const int dense = dense_mode ? 1 : 0;
  ug_check(is64(param) ? ugrid_total_variation_add_grad_f64(dp(param), dpm(grad), wx, wy, wz, dense, param.size(2), param.size(3), param.size(4),
                                                            param.numel(), ug_stream())
                       : ugrid_total_variation_add_grad(fp(param), fpm(grad), wx, wy, wz, dense, param.size(2), param.size(3), param.size(4),
                                                        param.numel(), ug_stream()),
           "total_variation_add_grad");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("total_variation_add_grad", &total_variation_add_grad, "Add total variation grad");
}
