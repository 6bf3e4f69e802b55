// ub360_utils_cuda for MI355X (FourierGrid/cuda/ub360_utils.cpp:19-22) + the segment_cumsum the reference's DistortionLoss
// calls (FourierGrid_model.py:689) but its extension never exported.
#include "ugrid_binding_common.h"

torch::Tensor cumdist_thres(torch::Tensor dist, float thres) {
  CHECK_INPUT(dist); CHECK_REAL(dist);
  TORCH_CHECK(dist.dim() == 2, "dist must be [n_rays, n_pts]");
  UG_GUARD(dist);
  auto mask = torch::empty({dist.size(0), dist.size(1)}, dist.options().dtype(at::kBool));
  ug_check((is64(dist) ? ugrid_cumdist_thres_f64(dp(dist), thres, dist.size(0), dist.size(1), (uint8_t *)mask.data_ptr<bool>(), ug_stream()) : ugrid_cumdist_thres(fp(dist), thres, dist.size(0), dist.size(1), (uint8_t *)mask.data_ptr<bool>(), ug_stream())), "cumdist_thres");
  return mask;
}

// (w_prefix, w_total, ws_prefix, ws_total); n_rays = ray_id[-1] + 1 (one host read, where the reference does ray_id.max()+1)
std::vector<torch::Tensor> segment_cumsum(torch::Tensor w, torch::Tensor s, torch::Tensor ray_id) {
  CHECK_INPUT(w); CHECK_INPUT(s); CHECK_INPUT(ray_id); CHECK_F32(w); CHECK_F32(s);
  TORCH_CHECK(ray_id.scalar_type() == at::kLong && w.dim() == 1 && w.sizes() == s.sizes() && w.sizes() == ray_id.sizes(),
              "w, s [n] float32 and ray_id [n] int64 expected");
  UG_GUARD(w);
  const int64_t n = w.numel();
  const int64_t n_rays = n > 0 ? ray_id[n - 1].item<int64_t>() + 1 : 0;
  auto w_prefix = torch::empty_like(w), ws_prefix = torch::empty_like(w);
  auto w_total = torch::zeros({n_rays}, w.options()), ws_total = torch::zeros({n_rays}, w.options());
  if (n_rays > 0) {
    auto scratch = torch::empty({2 * n_rays}, ray_id.options());
    ug_check(ugrid_segment_cumsum(fp(w), fp(s), ray_id.data_ptr<int64_t>(), n, n_rays, fpm(w_prefix), fpm(w_total), fpm(ws_prefix),
                                  fpm(ws_total), scratch.data_ptr<int64_t>(), ug_stream()), "segment_cumsum");
  }
  return {w_prefix, w_total, ws_prefix, ws_total};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("cumdist_thres", &cumdist_thres, "Generate mask for cumulative dist.");
  m.def("segment_cumsum", &segment_cumsum, "Exclusive per-ray running sums of w and w*s (DistortionLoss)");
}
