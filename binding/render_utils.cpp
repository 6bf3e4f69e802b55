// render_utils_cuda for MI355X: binds libugrid_hip.so under the 13 names of FourierGrid/cuda/render_utils.cpp:168-184.
#include "ugrid_binding_common.h"

std::vector<torch::Tensor> infer_t_minmax(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor xyz_min, torch::Tensor xyz_max,
                                          const float near, const float far) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(xyz_min); CHECK_INPUT(xyz_max);
  CHECK_F32(rays_o); CHECK_F32(rays_d); CHECK_F32(xyz_min); CHECK_F32(xyz_max);
  UG_GUARD(rays_o);
  const int64_t n = rays_o.size(0);
  auto t_min = torch::empty({n}, rays_o.options()), t_max = torch::empty({n}, rays_o.options());
  ug_check(ugrid_infer_t_minmax(fp(rays_o), fp(rays_d), fp(xyz_min), fp(xyz_max), near, far, n, fpm(t_min), fpm(t_max), ug_stream()),
           "infer_t_minmax");
  return {t_min, t_max};
}

torch::Tensor infer_n_samples(torch::Tensor rays_d, torch::Tensor t_min, torch::Tensor t_max, const float stepdist) {
  CHECK_INPUT(rays_d); CHECK_INPUT(t_min); CHECK_INPUT(t_max);
  CHECK_F32(rays_d); CHECK_F32(t_min); CHECK_F32(t_max);
  UG_GUARD(t_min);
  const int64_t n = t_min.size(0);
  auto out = torch::empty({n}, t_min.options().dtype(at::kLong));
  ug_check(ugrid_infer_n_samples(fp(rays_d), fp(t_min), fp(t_max), stepdist, n, out.data_ptr<int64_t>(), ug_stream()), "infer_n_samples");
  return out;
}

std::vector<torch::Tensor> infer_ray_start_dir(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor t_min) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(t_min);
  CHECK_F32(rays_o); CHECK_F32(rays_d); CHECK_F32(t_min);
  UG_GUARD(rays_o);
  auto start = torch::empty_like(rays_o), dirs = torch::empty_like(rays_o);
  ug_check(ugrid_infer_ray_start_dir(fp(rays_o), fp(rays_d), fp(t_min), rays_o.size(0), fpm(start), fpm(dirs), ug_stream()),
           "infer_ray_start_dir");
  return {start, dirs};
}

// -> {rays_pts [M,3], mask_outbbox bool [M], ray_id i64 [M], step_id i64 [M], N_steps i64 [R], t_min [R], t_max [R]}
// ONE host read of the sample total between the two halves, where the reference does N_steps.sum().item<int>()
// (render_utils_kernel.cu:212)
std::vector<torch::Tensor> sample_pts_on_rays(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor xyz_min, torch::Tensor xyz_max,
                                              const float near, const float far, const float stepdist) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(xyz_min); CHECK_INPUT(xyz_max);
  CHECK_F32(rays_o); CHECK_F32(rays_d); CHECK_F32(xyz_min); CHECK_F32(xyz_max);
  UG_GUARD(rays_o);
  const int64_t n = rays_o.size(0);
  const auto fo = rays_o.options().dtype(at::kFloat), lo = rays_o.options().dtype(at::kLong);
  auto t_min = torch::empty({n}, fo), t_max = torch::empty({n}, fo);
  auto n_steps = torch::empty({n}, lo), cumsum = torch::empty({n}, lo), total_d = torch::zeros({1}, lo);
  const int64_t ws_bytes = ugrid_scan_ws_bytes(n);
  auto ws = torch::empty({ws_bytes > 1 ? ws_bytes : 1}, rays_o.options().dtype(at::kByte));
  ug_check(ugrid_sample_pts_on_rays_count(fp(rays_o), fp(rays_d), fp(xyz_min), fp(xyz_max), near, far, stepdist, n, fpm(t_min), fpm(t_max),
                                          n_steps.data_ptr<int64_t>(), cumsum.data_ptr<int64_t>(), total_d.data_ptr<int64_t>(),
                                          ws.data_ptr(), ug_stream()), "sample_pts_on_rays (count)");
  const int64_t total = total_d.item<int64_t>();
  auto pts = torch::empty({total, 3}, fo);
  auto mask = torch::empty({total}, rays_o.options().dtype(at::kBool));
  auto ray_id = torch::empty({total}, lo), step_id = torch::empty({total}, lo);
  ug_check(ugrid_sample_pts_on_rays_fill(fp(rays_o), fp(rays_d), fp(xyz_min), fp(xyz_max), fp(t_min), cumsum.data_ptr<int64_t>(), stepdist, n,
                                         total, fpm(pts), (uint8_t *)mask.data_ptr<bool>(), ray_id.data_ptr<int64_t>(),
                                         step_id.data_ptr<int64_t>(), ug_stream()), "sample_pts_on_rays (fill)");
  return {pts, mask, ray_id, step_id, n_steps, t_min, t_max};
}

std::vector<torch::Tensor> sample_ndc_pts_on_rays(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor xyz_min, torch::Tensor xyz_max,
                                                  const int N_samples) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(xyz_min); CHECK_INPUT(xyz_max);
  CHECK_F32(rays_o); CHECK_F32(rays_d); CHECK_F32(xyz_min); CHECK_F32(xyz_max);
  UG_GUARD(rays_o);
  const int64_t n = rays_o.size(0);
  auto pts = torch::empty({n, N_samples, 3}, rays_o.options());
  auto mask = torch::empty({n, N_samples}, rays_o.options().dtype(at::kBool));
  ug_check(ugrid_sample_ndc_pts_on_rays(fp(rays_o), fp(rays_d), fp(xyz_min), fp(xyz_max), N_samples, n, fpm(pts),
                                        (uint8_t *)mask.data_ptr<bool>(), ug_stream()), "sample_ndc_pts_on_rays");
  return {pts, mask};
}

torch::Tensor sample_bg_pts_on_rays(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor t_max, const float bg_preserve,
                                    const int N_samples) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(t_max);
  CHECK_F32(rays_o); CHECK_F32(rays_d); CHECK_F32(t_max);
  UG_GUARD(rays_o);
  const int64_t n = rays_o.size(0);
  auto pts = torch::empty({n, N_samples, 3}, rays_o.options());
  ug_check(ugrid_sample_bg_pts_on_rays(fp(rays_o), fp(rays_d), fp(t_max), bg_preserve, N_samples, n, fpm(pts), ug_stream()),
           "sample_bg_pts_on_rays");
  return pts;
}

torch::Tensor maskcache_lookup(torch::Tensor world, torch::Tensor xyz, torch::Tensor xyz2ijk_scale, torch::Tensor xyz2ijk_shift) {
  CHECK_INPUT(world); CHECK_INPUT(xyz); CHECK_INPUT(xyz2ijk_scale); CHECK_INPUT(xyz2ijk_shift);
  CHECK_F32(xyz); CHECK_F32(xyz2ijk_scale); CHECK_F32(xyz2ijk_shift);
  TORCH_CHECK(world.scalar_type() == at::kBool && world.dim() == 3, "world must be a 3-D bool tensor");
  UG_GUARD(xyz);
  const int64_t n = xyz.size(0);
  auto out = torch::empty({n}, xyz.options().dtype(at::kBool));
  ug_check(ugrid_maskcache_lookup((const uint8_t *)world.data_ptr<bool>(), fp(xyz), fp(xyz2ijk_scale), fp(xyz2ijk_shift), world.size(0),
                                  world.size(1), world.size(2), n, (uint8_t *)out.data_ptr<bool>(), ug_stream()), "maskcache_lookup");
  return out;
}

std::vector<torch::Tensor> raw2alpha(torch::Tensor density, const float shift, const float interval) {
  CHECK_INPUT(density); CHECK_F32(density);
  UG_GUARD(density);
  auto exp_d = torch::empty_like(density), alpha = torch::empty_like(density);
  ug_check(ugrid_raw2alpha(fp(density), shift, interval, nullptr, density.size(0), fpm(exp_d), fpm(alpha), ug_stream()), "raw2alpha");
  return {exp_d, alpha};
}

std::vector<torch::Tensor> raw2alpha_nonuni(torch::Tensor density, const float shift, torch::Tensor interval) {
  CHECK_INPUT(density); CHECK_INPUT(interval); CHECK_F32(density); CHECK_F32(interval);
  UG_GUARD(density);
  auto exp_d = torch::empty_like(density), alpha = torch::empty_like(density);
  ug_check(ugrid_raw2alpha(fp(density), shift, 0.f, fp(interval), density.size(0), fpm(exp_d), fpm(alpha), ug_stream()), "raw2alpha_nonuni");
  return {exp_d, alpha};
}

torch::Tensor raw2alpha_backward(torch::Tensor exp, torch::Tensor grad_back, const float interval) {
  CHECK_INPUT(exp); CHECK_INPUT(grad_back); CHECK_F32(exp); CHECK_F32(grad_back);
  UG_GUARD(exp);
  auto grad = torch::empty_like(exp);
  ug_check(ugrid_raw2alpha_backward(fp(exp), fp(grad_back), interval, nullptr, exp.size(0), fpm(grad), ug_stream()), "raw2alpha_backward");
  return grad;
}

torch::Tensor raw2alpha_nonuni_backward(torch::Tensor exp, torch::Tensor grad_back, torch::Tensor interval) {
  CHECK_INPUT(exp); CHECK_INPUT(grad_back); CHECK_INPUT(interval); CHECK_F32(exp); CHECK_F32(grad_back); CHECK_F32(interval);
  UG_GUARD(exp);
  auto grad = torch::empty_like(exp);
  ug_check(ugrid_raw2alpha_backward(fp(exp), fp(grad_back), 0.f, fp(interval), exp.size(0), fpm(grad), ug_stream()),
           "raw2alpha_nonuni_backward");
  return grad;
}

// -> {weight [n], T [n], alphainv_last [R], i_start i64 [R], i_end i64 [R]} (render_utils_kernel.cu:650); no host sync
std::vector<torch::Tensor> alpha2weight(torch::Tensor alpha, torch::Tensor ray_id, const int n_rays) {
  CHECK_INPUT(alpha); CHECK_INPUT(ray_id); CHECK_F32(alpha);
  TORCH_CHECK(ray_id.scalar_type() == at::kLong, "ray_id must be int64");
  TORCH_CHECK(ray_id.numel() >= alpha.size(0), "ray_id has fewer entries than alpha.size(0)");
  UG_GUARD(alpha);
  // the kernel writes all n = alpha.size(0) entries; for a [R,S] alpha (fast_color_thres == 0) the reference leaves the rest of
  // its zeros_like / ones_like outputs untouched (render_utils_kernel.cu:620-626): same here
  auto weight = alpha.dim() == 1 ? torch::empty_like(alpha) : torch::zeros_like(alpha);
  auto T = alpha.dim() == 1 ? torch::empty_like(alpha) : torch::ones_like(alpha);
  auto last = torch::empty({n_rays}, alpha.options());
  auto i_start = torch::empty({n_rays}, ray_id.options()), i_end = torch::empty({n_rays}, ray_id.options());
  ug_check(ugrid_alpha2weight(fp(alpha), ray_id.data_ptr<int64_t>(), alpha.size(0), n_rays, fpm(weight), fpm(T), fpm(last),
                              i_start.data_ptr<int64_t>(), i_end.data_ptr<int64_t>(), ug_stream()), "alpha2weight");
  return {weight, T, last, i_start, i_end};
}

torch::Tensor alpha2weight_backward(torch::Tensor alpha, torch::Tensor weight, torch::Tensor T, torch::Tensor alphainv_last,
                                    torch::Tensor i_start, torch::Tensor i_end, const int n_rays, torch::Tensor grad_weights,
                                    torch::Tensor grad_last) {
  CHECK_INPUT(alpha); CHECK_INPUT(weight); CHECK_INPUT(T); CHECK_INPUT(alphainv_last); CHECK_INPUT(i_start); CHECK_INPUT(i_end);
  CHECK_INPUT(grad_weights); CHECK_INPUT(grad_last);
  CHECK_F32(alpha); CHECK_F32(weight); CHECK_F32(T); CHECK_F32(alphainv_last); CHECK_F32(grad_weights); CHECK_F32(grad_last);
  UG_GUARD(alpha);
  auto grad = torch::empty_like(alpha);
  ug_check(ugrid_alpha2weight_backward(fp(alpha), fp(weight), fp(T), fp(alphainv_last), i_start.data_ptr<int64_t>(), i_end.data_ptr<int64_t>(),
                                       alpha.size(0), n_rays, fp(grad_weights), fp(grad_last), fpm(grad), ug_stream()),
           "alpha2weight_backward");
  return grad;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("infer_t_minmax", &infer_t_minmax, "Inference t_min and t_max of ray-bbox intersection");
  m.def("infer_n_samples", &infer_n_samples, "Inference the number of points to sample on each ray");
  m.def("infer_ray_start_dir", &infer_ray_start_dir, "Inference the starting point and shooting direction of each ray");
  m.def("sample_pts_on_rays", &sample_pts_on_rays, "Sample points on rays");
  m.def("sample_ndc_pts_on_rays", &sample_ndc_pts_on_rays, "Sample points on rays");
  m.def("sample_bg_pts_on_rays", &sample_bg_pts_on_rays, "Sample points on bg");
  m.def("maskcache_lookup", &maskcache_lookup, "Lookup to skip know freespace.");
  m.def("raw2alpha", &raw2alpha, "Raw values [-inf, inf] to alpha [0, 1].");
  m.def("raw2alpha_backward", &raw2alpha_backward, "Backward pass of the raw to alpha");
  m.def("raw2alpha_nonuni", &raw2alpha_nonuni, "Raw values [-inf, inf] to alpha [0, 1].");
  m.def("raw2alpha_nonuni_backward", &raw2alpha_nonuni_backward, "Backward pass of the raw to alpha");
  m.def("alpha2weight", &alpha2weight, "Per-point alpha to accumulated blending weight");
  m.def("alpha2weight_backward", &alpha2weight_backward, "Backward pass of alpha2weight");
}
