// adam_upd_cuda for MI355X (FourierGrid/cuda/adam_upd.cpp:77-87): param / exp_avg / exp_avg_sq are updated in place.
#include "ugrid_binding_common.h"

static void run(torch::Tensor &param, torch::Tensor &grad, torch::Tensor &exp_avg, torch::Tensor &exp_avg_sq, const torch::Tensor *perlr,
                int step, float beta1, float beta2, float lr, float eps, int mode, const char *what) {
  CHECK_INPUT(param); CHECK_INPUT(grad); CHECK_INPUT(exp_avg); CHECK_INPUT(exp_avg_sq);
  CHECK_REAL(param); CHECK_SAME(grad, param); CHECK_SAME(exp_avg, param); CHECK_SAME(exp_avg_sq, param);
  if (perlr) { CHECK_CUDA((*perlr)); CHECK_CONTIGUOUS((*perlr)); CHECK_SAME((*perlr), param); }
  UG_GUARD(param);
  ug_check(is64(param) ? ugrid_adam_upd_f64(dpm(param), dp(grad), dpm(exp_avg), dpm(exp_avg_sq), perlr ? dp(*perlr) : nullptr, param.numel(), step,
                                            beta1, beta2, lr, eps, mode, ug_stream())
                       : ugrid_adam_upd(fpm(param), fp(grad), fpm(exp_avg), fpm(exp_avg_sq), perlr ? fp(*perlr) : nullptr, param.numel(), step,
                                        beta1, beta2, lr, eps, mode, ug_stream()),
           what);
}

void adam_upd(torch::Tensor param, torch::Tensor grad, torch::Tensor exp_avg, torch::Tensor exp_avg_sq, int step, float beta1, float beta2,
              float lr, float eps) {
  run(param, grad, exp_avg, exp_avg_sq, nullptr, step, beta1, beta2, lr, eps, 0, "adam_upd");
}
void masked_adam_upd(torch::Tensor param, torch::Tensor grad, torch::Tensor exp_avg, torch::Tensor exp_avg_sq, int step, float beta1,
                     float beta2, float lr, float eps) {
  run(param, grad, exp_avg, exp_avg_sq, nullptr, step, beta1, beta2, lr, eps, 1, "masked_adam_upd");
}
void adam_upd_with_perlr(torch::Tensor param, torch::Tensor grad, torch::Tensor exp_avg, torch::Tensor exp_avg_sq, torch::Tensor perlr,
                         int step, float beta1, float beta2, float lr, float eps) {
  run(param, grad, exp_avg, exp_avg_sq, &perlr, step, beta1, beta2, lr, eps, 2, "adam_upd_with_perlr");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("adam_upd", &adam_upd, "Adam update");
  m.def("masked_adam_upd", &masked_adam_upd, "Adam update ignoring zero grad");
  m.def("adam_upd_with_perlr", &adam_upd_with_perlr, "Adam update ignoring zero grad with per-voxel lr");
}
