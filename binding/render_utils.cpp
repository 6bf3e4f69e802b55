// render_utils_cuda for MI355X: binds libugrid_hip.so under the 13 names of FourierGrid/cuda/render_utils.cpp:168-184.
// float tensors go to the entry points of ugrid_hip.h, double tensors to their twins of ugrid_hip_f64.h (the reference dispatches both).
#include "ugrid_binding_common.h"

std::vector<torch::Tensor> infer_t_minmax(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor xyz_min, torch::Tensor xyz_max,
                                          const float near, const float far) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(xyz_min); CHECK_INPUT(xyz_max);
  CHECK_REAL(rays_o); CHECK_SAME(rays_d, rays_o); CHECK_SAME(xyz_min, rays_o); CHECK_SAME(xyz_max, rays_o);
  UG_GUARD(rays_o);
  const int64_t n = rays_o.size(0);
  auto t_min = torch::empty({n}, rays_o.options()), t_max = torch::empty({n}, rays_o.options());
  ug_check((is64(rays_o) ? ugrid_infer_t_minmax_f64(dp(rays_o), dp(rays_d), dp(xyz_min), dp(xyz_max), near, far, n, dpm(t_min), dpm(t_max), ug_stream()) : ugrid_infer_t_minmax(fp(rays_o), fp(rays_d), fp(xyz_min), fp(xyz_max), near, far, n, fpm(t_min), fpm(t_max), ug_stream())),
           "infer_t_minmax");
  return {t_min, t_max};
}

torch::Tensor infer_n_samples(torch::Tensor rays_d, torch::Tensor t_min, torch::Tensor t_max, const float stepdist) {
  CHECK_INPUT(rays_d); CHECK_INPUT(t_min); CHECK_INPUT(t_max);
  CHECK_REAL(rays_d); CHECK_SAME(t_min, rays_d); CHECK_SAME(t_max, rays_d);
  UG_GUARD(t_min);
  const int64_t n = t_min.size(0);
  auto out = torch::empty({n}, t_min.options().dtype(at::kLong));
  ug_check((is64(t_min) ? ugrid_infer_n_samples_f64(dp(rays_d), dp(t_min), dp(t_max), stepdist, n, out.data_ptr<int64_t>(), ug_stream()) : ugrid_infer_n_samples(fp(rays_d), fp(t_min), fp(t_max), stepdist, n, out.data_ptr<int64_t>(), ug_stream())), "infer_n_samples");
  return out;
}

std::vector<torch::Tensor> infer_ray_start_dir(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor t_min) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(t_min);
  CHECK_REAL(rays_o); CHECK_SAME(rays_d, rays_o); CHECK_SAME(t_min, rays_o);
  UG_GUARD(rays_o);
  auto start = torch::empty_like(rays_o), dirs = torch::empty_like(rays_o);
  ug_check((is64(rays_o) ? ugrid_infer_ray_start_dir_f64(dp(rays_o), dp(rays_d), dp(t_min), rays_o.size(0), dpm(start), dpm(dirs), ug_stream()) : ugrid_infer_ray_start_dir(fp(rays_o), fp(rays_d), fp(t_min), rays_o.size(0), fpm(start), fpm(dirs), ug_stream())),
           "infer_ray_start_dir");
  return {start, dirs};
}

// -> {rays_pts [M,3], mask_outbbox bool [M], ray_id i64 [M], step_id i64 [M], N_steps i64 [R], t_min [R], t_max [R]}
// ONE host read of the sample total between the two halves, where the reference does N_steps.sum().item<int>()
// (render_utils_kernel.cu:212)
std::vector<torch::Tensor> sample_pts_on_rays(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor xyz_min, torch::Tensor xyz_max,
                                              const float near, const float far, const float stepdist) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(xyz_min); CHECK_INPUT(xyz_max);
  CHECK_REAL(rays_o); CHECK_SAME(rays_d, rays_o); CHECK_SAME(xyz_min, rays_o); CHECK_SAME(xyz_max, rays_o);
  UG_GUARD(rays_o);
  const int64_t n = rays_o.size(0);
  const auto fo = rays_o.options(), lo = rays_o.options().dtype(at::kLong);      // (float or double, as the rays)
  auto t_min = torch::empty({n}, fo), t_max = torch::empty({n}, fo);
  auto n_steps = torch::empty({n}, lo), cumsum = torch::empty({n}, lo), total_d = torch::zeros({1}, lo);
  const int64_t ws_bytes = ugrid_scan_ws_bytes(n);
  auto ws = torch::empty({ws_bytes > 1 ? ws_bytes : 1}, rays_o.options().dtype(at::kByte));
  ug_check((is64(rays_o) ? ugrid_sample_pts_on_rays_count_f64(dp(rays_o), dp(rays_d), dp(xyz_min), dp(xyz_max), near, far, stepdist, n, dpm(t_min), dpm(t_max),
                                          n_steps.data_ptr<int64_t>(), cumsum.data_ptr<int64_t>(), total_d.data_ptr<int64_t>(), ug_stream()) : ugrid_sample_pts_on_rays_count(fp(rays_o), fp(rays_d), fp(xyz_min), fp(xyz_max), near, far, stepdist, n, fpm(t_min), fpm(t_max),
                                          n_steps.data_ptr<int64_t>(), cumsum.data_ptr<int64_t>(), total_d.data_ptr<int64_t>(),
                                          ws.data_ptr(), ug_stream())), "sample_pts_on_rays (count)");
  const int64_t total = total_d.item<int64_t>();
  auto pts = torch::empty({total, 3}, fo);
  auto mask = torch::empty({total}, rays_o.options().dtype(at::kBool));
  auto ray_id = torch::empty({total}, lo), step_id = torch::empty({total}, lo);
  ug_check((is64(rays_o) ? ugrid_sample_pts_on_rays_fill_f64(dp(rays_o), dp(rays_d), dp(xyz_min), dp(xyz_max), dp(t_min), cumsum.data_ptr<int64_t>(), stepdist, n,
                                         total, dpm(pts), (uint8_t *)mask.data_ptr<bool>(), ray_id.data_ptr<int64_t>(),
                                         step_id.data_ptr<int64_t>(), ug_stream()) : ugrid_sample_pts_on_rays_fill(fp(rays_o), fp(rays_d), fp(xyz_min), fp(xyz_max), fp(t_min), cumsum.data_ptr<int64_t>(), stepdist, n,
                                         total, fpm(pts), (uint8_t *)mask.data_ptr<bool>(), ray_id.data_ptr<int64_t>(),
                                         step_id.data_ptr<int64_t>(), ug_stream())), "sample_pts_on_rays (fill)");
  return {pts, mask, ray_id, step_id, n_steps, t_min, t_max};
}

std::vector<torch::Tensor> sample_ndc_pts_on_rays(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor xyz_min, torch::Tensor xyz_max,
                                                  const int N_samples) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(xyz_min); CHECK_INPUT(xyz_max);
  CHECK_REAL(rays_o); CHECK_SAME(rays_d, rays_o); CHECK_SAME(xyz_min, rays_o); CHECK_SAME(xyz_max, rays_o);
  UG_GUARD(rays_o);
  const int64_t n = rays_o.size(0);
  auto pts = torch::empty({n, N_samples, 3}, rays_o.options());
  auto mask = torch::empty({n, N_samples}, rays_o.options().dtype(at::kBool));
  ug_check((is64(rays_o) ? ugrid_sample_ndc_pts_on_rays_f64(dp(rays_o), dp(rays_d), dp(xyz_min), dp(xyz_max), N_samples, n, dpm(pts),
                                        (uint8_t *)mask.data_ptr<bool>(), ug_stream()) : ugrid_sample_ndc_pts_on_rays(fp(rays_o), fp(rays_d), fp(xyz_min), fp(xyz_max), N_samples, n, fpm(pts),
                                        (uint8_t *)mask.data_ptr<bool>(), ug_stream())), "sample_ndc_pts_on_rays");
  return {pts, mask};
}

torch::Tensor sample_bg_pts_on_rays(torch::Tensor rays_o, torch::Tensor rays_d, torch::Tensor t_max, const float bg_preserve,
                                    const int N_samples) {
  CHECK_INPUT(rays_o); CHECK_INPUT(rays_d); CHECK_INPUT(t_max);
  CHECK_REAL(rays_o); CHECK_SAME(rays_d, rays_o); CHECK_SAME(t_max, rays_o);
  UG_GUARD(rays_o);
  const int64_t n = rays_o.size(0);
  auto pts = torch::empty({n, N_samples, 3}, rays_o.options());
  ug_check((is64(rays_o) ? ugrid_sample_bg_pts_on_rays_f64(dp(rays_o), dp(rays_d), dp(t_max), bg_preserve, N_samples, n, dpm(pts), ug_stream()) : ugrid_sample_bg_pts_on_rays(fp(rays_o), fp(rays_d), fp(t_max), bg_preserve, N_samples, n, fpm(pts), ug_stream())),
           "sample_bg_pts_on_rays");
  return pts;
}

torch::Tensor maskcache_lookup(torch::Tensor world, torch::Tensor xyz, torch::Tensor xyz2ijk_scale, torch::Tensor xyz2ijk_shift) {
  CHECK_INPUT(world); CHECK_INPUT(xyz); CHECK_INPUT(xyz2ijk_scale); CHECK_INPUT(xyz2ijk_shift);
  CHECK_REAL(xyz); CHECK_SAME(xyz2ijk_scale, xyz); CHECK_SAME(xyz2ijk_shift, xyz);
  TORCH_CHECK(world.scalar_type() == at::kBool && world.dim() == 3, "world must be a 3-D bool tensor");
  UG_GUARD(xyz);
  const int64_t n = xyz.size(0);
  auto out = torch::empty({n}, xyz.options().dtype(at::kBool));
  ug_check((is64(xyz) ? ugrid_maskcache_lookup_f64((const uint8_t *)world.data_ptr<bool>(), dp(xyz), dp(xyz2ijk_scale), dp(xyz2ijk_shift), world.size(0),
                                  world.size(1), world.size(2), n, (uint8_t *)out.data_ptr<bool>(), ug_stream()) : ugrid_maskcache_lookup((const uint8_t *)world.data_ptr<bool>(), fp(xyz), fp(xyz2ijk_scale), fp(xyz2ijk_shift), world.size(0),
                                  world.size(1), world.size(2), n, (uint8_t *)out.data_ptr<bool>(), ug_stream())), "maskcache_lookup");
  return out;
}

std::vector<torch::Tensor> raw2alpha(torch::Tensor density, const float shift, const float interval) {
  CHECK_INPUT(density); CHECK_REAL(density);
  UG_GUARD(density);
  auto exp_d = torch::empty_like(density), alpha = torch::empty_like(density);
  ug_check((is64(density) ? ugrid_raw2alpha_f64(dp(density), shift, interval, nullptr, density.size(0), dpm(exp_d), dpm(alpha), ug_stream()) : ugrid_raw2alpha(fp(density), shift, interval, nullptr, density.size(0), fpm(exp_d), fpm(alpha), ug_stream())), "raw2alpha");
  return {exp_d, alpha};
}

std::vector<torch::Tensor> raw2alpha_nonuni(torch::Tensor density, const float shift, torch::Tensor interval) {
  CHECK_INPUT(density); CHECK_INPUT(interval); CHECK_REAL(density); CHECK_SAME(interval, density);
  UG_GUARD(density);
  auto exp_d = torch::empty_like(density), alpha = torch::empty_like(density);
  ug_check((is64(density) ? ugrid_raw2alpha_f64(dp(density), shift, 0.f, dp(interval), density.size(0), dpm(exp_d), dpm(alpha), ug_stream()) : ugrid_raw2alpha(fp(density), shift, 0.f, fp(interval), density.size(0), fpm(exp_d), fpm(alpha), ug_stream())), "raw2alpha_nonuni");
  return {exp_d, alpha};
}

torch::Tensor raw2alpha_backward(torch::Tensor exp, torch::Tensor grad_back, const float interval) {
  CHECK_INPUT(exp); CHECK_INPUT(grad_back); CHECK_REAL(exp); CHECK_SAME(grad_back, exp);
  UG_GUARD(exp);
  auto grad = torch::empty_like(exp);
  ug_check((is64(exp) ? ugrid_raw2alpha_backward_f64(dp(exp), dp(grad_back), interval, nullptr, exp.size(0), dpm(grad), ug_stream()) : ugrid_raw2alpha_backward(fp(exp), fp(grad_back), interval, nullptr, exp.size(0), fpm(grad), ug_stream())), "raw2alpha_backward");
  return grad;
}

torch::Tensor raw2alpha_nonuni_backward(torch::Tensor exp, torch::Tensor grad_back, torch::Tensor interval) {
  CHECK_INPUT(exp); CHECK_INPUT(grad_back); CHECK_INPUT(interval); CHECK_REAL(exp); CHECK_SAME(grad_back, exp); CHECK_SAME(interval, exp);
  UG_GUARD(exp);
  auto grad = torch::empty_like(exp);
  ug_check((is64(exp) ? ugrid_raw2alpha_backward_f64(dp(exp), dp(grad_back), 0.f, dp(interval), exp.size(0), dpm(grad), ug_stream()) : ugrid_raw2alpha_backward(fp(exp), fp(grad_back), 0.f, fp(interval), exp.size(0), fpm(grad), ug_stream())),
           "raw2alpha_nonuni_backward");
  return grad;
}

// -> {weight [n], T [n], alphainv_last [R], i_start i64 [R], i_end i64 [R]} (render_utils_kernel.cu:650); no host sync
std::vector<torch::Tensor> alpha2weight(torch::Tensor alpha, torch::Tensor ray_id, const int n_rays) {
  CHECK_INPUT(alpha); CHECK_INPUT(ray_id); CHECK_REAL(alpha);
  TORCH_CHECK(ray_id.scalar_type() == at::kLong, "ray_id must be int64");
  TORCH_CHECK(ray_id.numel() >= alpha.size(0), "ray_id has fewer entries than alpha.size(0)");
  UG_GUARD(alpha);
  // the kernel writes all n = alpha.size(0) entries; for a [R,S] alpha (fast_color_thres == 0) the reference leaves the rest of
  // its zeros_like / ones_like outputs untouched (render_utils_kernel.cu:620-626): same here
  auto weight = alpha.dim() == 1 ? torch::empty_like(alpha) : torch::zeros_like(alpha);
  auto T = alpha.dim() == 1 ? torch::empty_like(alpha) : torch::ones_like(alpha);
  auto last = torch::empty({n_rays}, alpha.options());
  auto i_start = torch::empty({n_rays}, ray_id.options()), i_end = torch::empty({n_rays}, ray_id.options());
  ug_check((is64(alpha) ? ugrid_alpha2weight_f64(dp(alpha), ray_id.data_ptr<int64_t>(), alpha.size(0), n_rays, dpm(weight), dpm(T), dpm(last),
                              i_start.data_ptr<int64_t>(), i_end.data_ptr<int64_t>(), ug_stream()) : ugrid_alpha2weight(fp(alpha), ray_id.data_ptr<int64_t>(), alpha.size(0), n_rays, fpm(weight), fpm(T), fpm(last),
                              i_start.data_ptr<int64_t>(), i_end.data_ptr<int64_t>(), ug_stream())), "alpha2weight");
  return {weight, T, last, i_start, i_end};
}

torch::Tensor alpha2weight_backward(torch::Tensor alpha, torch::Tensor weight, torch::Tensor T, torch::Tensor alphainv_last,
                                    torch::Tensor i_start, torch::Tensor i_end, const int n_rays, torch::Tensor grad_weights,
                                    torch::Tensor grad_last) {
  CHECK_INPUT(alpha); CHECK_INPUT(weight); CHECK_INPUT(T); CHECK_INPUT(alphainv_last); CHECK_INPUT(i_start); CHECK_INPUT(i_end);
  CHECK_INPUT(grad_weights); CHECK_INPUT(grad_last);
  CHECK_REAL(alpha); CHECK_SAME(weight, alpha); CHECK_SAME(T, alpha); CHECK_SAME(alphainv_last, alpha); CHECK_SAME(grad_weights, alpha); CHECK_SAME(grad_last, alpha);
  UG_GUARD(alpha);
  auto grad = torch::empty_like(alpha);
  ug_check((is64(alpha) ? ugrid_alpha2weight_backward_f64(dp(alpha), dp(weight), dp(T), dp(alphainv_last), i_start.data_ptr<int64_t>(), i_end.data_ptr<int64_t>(),
                                       alpha.size(0), n_rays, dp(grad_weights), dp(grad_last), dpm(grad), ug_stream()) : ugrid_alpha2weight_backward(fp(alpha), fp(weight), fp(T), fp(alphainv_last), i_start.data_ptr<int64_t>(), i_end.data_ptr<int64_t>(),
                                       alpha.size(0), n_rays, fp(grad_weights), fp(grad_last), fpm(grad), ug_stream())),
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
