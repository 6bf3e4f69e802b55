"""MI355X drop-in for the reference extension module `render_utils_cuda`
(/root/reference/FourierGrid/cuda/render_utils.cpp:170-184): same 13 function names, argument order,
return arity, dtypes (float32, or float64 through the twins of libugrid_hip_f64.so: the reference dispatches both) and output
initialisation; device work is done by libugrid_hip.so on the tensor's device and torch's current stream (the reference uses the legacy default stream and no
device guard -- SURVEY.md section 8b).

`float` arguments may be Python floats or 0-d / 1-element tensors (pybind's __float__ conversion in
the reference; a device tensor costs one host sync, exactly as there).
"""
import torch

from . import _lib

_L = _lib.load()
_p, _s = _lib.ptr, _lib.stream_of


def _f(x):
    return float(x)


def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    _lib.require_cuda(("rays_o", rays_o), ("rays_d", rays_d), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
    dt = _lib.real_dtype(("rays_o", rays_o), ("rays_d", rays_d), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
    n = rays_o.size(0)
    t_min = torch.empty(n, dtype=rays_o.dtype, device=rays_o.device)
    t_max = torch.empty_like(t_min)
    with _lib.guard(rays_o.device):
        _lib.check(_lib.entry("ugrid_infer_t_minmax", dt)(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), _f(near), _f(far),
                                           n, _p(t_min), _p(t_max), _s(rays_o)), "infer_t_minmax")
    return [t_min, t_max]


def infer_n_samples(rays_d, t_min, t_max, stepdist):
    _lib.require_cuda(("rays_d", rays_d), ("t_min", t_min), ("t_max", t_max))
    dt = _lib.real_dtype(("rays_d", rays_d), ("t_min", t_min), ("t_max", t_max))
    n = t_min.size(0)
    out = torch.empty(n, dtype=torch.int64, device=t_min.device)
    with _lib.guard(t_min.device):
        _lib.check(_lib.entry("ugrid_infer_n_samples", dt)(_p(rays_d), _p(t_min), _p(t_max), _f(stepdist), n, _p(out),
                                            _s(t_min)), "infer_n_samples")
    return out


def infer_ray_start_dir(rays_o, rays_d, t_min):
    _lib.require_cuda(("rays_o", rays_o), ("rays_d", rays_d), ("t_min", t_min))
    dt = _lib.real_dtype(("rays_o", rays_o), ("rays_d", rays_d), ("t_min", t_min))
    start = torch.empty_like(rays_o)
    dirs = torch.empty_like(rays_o)
    with _lib.guard(rays_o.device):
        _lib.check(_lib.entry("ugrid_infer_ray_start_dir", dt)(_p(rays_o), _p(rays_d), _p(t_min), rays_o.size(0), _p(start),
                                                _p(dirs), _s(rays_o)), "infer_ray_start_dir")
    return [start, dirs]


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """-> [rays_pts[M,3], mask_outbbox bool[M], ray_id i64[M], step_id i64[M], N_steps i64[R], t_min[R], t_max[R]].
    One host sync to read M, like the reference's N_steps.sum().item() (render_utils_kernel.cu:212)."""
    _lib.require_cuda(("rays_o", rays_o), ("rays_d", rays_d), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
    dt = _lib.real_dtype(("rays_o", rays_o), ("rays_d", rays_d), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
    dev = rays_o.device
    n = rays_o.size(0)
    t_min = torch.empty(n, dtype=dt, device=dev)
    t_max = torch.empty_like(t_min)
    n_steps = torch.empty(n, dtype=torch.int64, device=dev)
    cumsum = torch.empty_like(n_steps)
    total_d = torch.zeros(1, dtype=torch.int64, device=dev)
    f64 = dt == torch.float64
    ws = None if f64 else torch.empty(max(1, _L.ugrid_scan_ws_bytes(n)), dtype=torch.uint8, device=dev)
    with _lib.guard(dev):
        st = _s(rays_o)
        _lib.check(_lib.entry("ugrid_sample_pts_on_rays_count", dt)(
            _p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), _f(near), _f(far), _f(stepdist), n, _p(t_min), _p(t_max), _p(n_steps),
            _p(cumsum), _p(total_d), *(() if f64 else (_p(ws),)), st), "sample_pts_on_rays(count)")
        total = int(total_d.item())
        pts = torch.empty(total, 3, dtype=dt, device=dev)
        mask = torch.empty(total, dtype=torch.bool, device=dev)
        ray_id = torch.empty(total, dtype=torch.int64, device=dev)
        step_id = torch.empty(total, dtype=torch.int64, device=dev)
        _lib.check(_lib.entry("ugrid_sample_pts_on_rays_fill", dt)(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), _p(t_min),
                                                    _p(cumsum), _f(stepdist), n, total, _p(pts), _p(mask),
                                                    _p(ray_id), _p(step_id), st), "sample_pts_on_rays(fill)")
    return [pts, mask, ray_id, step_id, n_steps, t_min, t_max]


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    _lib.require_cuda(("rays_o", rays_o), ("rays_d", rays_d), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
    dt = _lib.real_dtype(("rays_o", rays_o), ("rays_d", rays_d), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
    n = rays_o.size(0)
    N_samples = int(N_samples)
    pts = torch.empty(n, N_samples, 3, dtype=dt, device=rays_o.device)
    mask = torch.empty(n, N_samples, dtype=torch.bool, device=rays_o.device)
    with _lib.guard(rays_o.device):
        _lib.check(_lib.entry("ugrid_sample_ndc_pts_on_rays", dt)(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), N_samples, n,
                                                   _p(pts), _p(mask), _s(rays_o)), "sample_ndc_pts_on_rays")
    return [pts, mask]


def sample_bg_pts_on_rays(rays_o, rays_d, t_max, bg_preserve, N_samples):
    _lib.require_cuda(("rays_o", rays_o), ("rays_d", rays_d), ("t_max", t_max))
    dt = _lib.real_dtype(("rays_o", rays_o), ("rays_d", rays_d), ("t_max", t_max))
    n = rays_o.size(0)
    N_samples = int(N_samples)
    pts = torch.empty(n, N_samples, 3, dtype=dt, device=rays_o.device)
    with _lib.guard(rays_o.device):
        _lib.check(_lib.entry("ugrid_sample_bg_pts_on_rays", dt)(_p(rays_o), _p(rays_d), _p(t_max), _f(bg_preserve), N_samples, n,
                                                  _p(pts), _s(rays_o)), "sample_bg_pts_on_rays")
    return pts


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    _lib.require_cuda(("world", world), ("xyz", xyz), ("xyz2ijk_scale", xyz2ijk_scale),
                      ("xyz2ijk_shift", xyz2ijk_shift))
    dt = _lib.real_dtype(("xyz", xyz), ("xyz2ijk_scale", xyz2ijk_scale), ("xyz2ijk_shift", xyz2ijk_shift))
    if world.dtype != torch.bool or world.dim() != 3:
        raise RuntimeError("world must be a 3-D bool tensor")
    n = xyz.size(0)
    out = torch.empty(n, dtype=torch.bool, device=xyz.device)
    with _lib.guard(xyz.device):
        _lib.check(_lib.entry("ugrid_maskcache_lookup", dt)(_p(world), _p(xyz), _p(xyz2ijk_scale), _p(xyz2ijk_shift),
                                             world.size(0), world.size(1), world.size(2), n, _p(out), _s(xyz)),
                   "maskcache_lookup")
    return out


def raw2alpha(density, shift, interval):
    _lib.require_cuda(("density", density))
    dt = _lib.real_dtype(("density", density))
    exp_d = torch.empty_like(density)
    alpha = torch.empty_like(density)
    with _lib.guard(density.device):
        _lib.check(_lib.entry("ugrid_raw2alpha", dt)(_p(density), _f(shift), _f(interval), None, density.size(0), _p(exp_d),
                                      _p(alpha), _s(density)), "raw2alpha")
    return [exp_d, alpha]


def raw2alpha_nonuni(density, shift, interval):
    _lib.require_cuda(("density", density), ("interval", interval))
    dt = _lib.real_dtype(("density", density), ("interval", interval))
    exp_d = torch.empty_like(density)
    alpha = torch.empty_like(density)
    with _lib.guard(density.device):
        _lib.check(_lib.entry("ugrid_raw2alpha", dt)(_p(density), _f(shift), 0.0, _p(interval), density.size(0), _p(exp_d),
                                      _p(alpha), _s(density)), "raw2alpha_nonuni")
    return [exp_d, alpha]


def raw2alpha_backward(exp, grad_back, interval):
    _lib.require_cuda(("exp", exp), ("grad_back", grad_back))
    dt = _lib.real_dtype(("exp", exp), ("grad_back", grad_back))
    grad = torch.empty_like(exp)
    with _lib.guard(exp.device):
        _lib.check(_lib.entry("ugrid_raw2alpha_backward", dt)(_p(exp), _p(grad_back), _f(interval), None, exp.size(0), _p(grad),
                                               _s(exp)), "raw2alpha_backward")
    return grad


def raw2alpha_nonuni_backward(exp, grad_back, interval):
    _lib.require_cuda(("exp", exp), ("grad_back", grad_back), ("interval", interval))
    dt = _lib.real_dtype(("exp", exp), ("grad_back", grad_back), ("interval", interval))
    grad = torch.empty_like(exp)
    with _lib.guard(exp.device):
        _lib.check(_lib.entry("ugrid_raw2alpha_backward", dt)(_p(exp), _p(grad_back), 0.0, _p(interval), exp.size(0), _p(grad),
                                               _s(exp)), "raw2alpha_nonuni_backward")
    return grad


def alpha2weight(alpha, ray_id, n_rays):
    """-> [weight[n], T[n], alphainv_last[R], i_start i64[R], i_end i64[R]]; no host sync (the reference
    does `i_end[ray_id[n-1]] = n` through host indexing, render_utils_kernel.cu:635)."""
    _lib.require_cuda(("alpha", alpha), ("ray_id", ray_id))
    dt = _lib.real_dtype(("alpha", alpha))
    if ray_id.dtype != torch.int64:
        raise RuntimeError("ray_id must be int64")
    n, n_rays = alpha.size(0), int(n_rays)
    dev = alpha.device
    if ray_id.numel() < n:
        raise RuntimeError("ray_id has fewer entries than alpha.size(0)")
    if alpha.dim() == 1:
        weight = torch.empty_like(alpha)      # the kernel writes all n entries (defaults 0 / 1 after the early stop)
        T = torch.empty_like(alpha)
    else:
        # the reference takes n_pts = alpha.size(0) whatever the rank (render_utils_kernel.cu:620-626) and leaves the
        # rest of its zeros_like / ones_like outputs untouched: same here for a [R,S] alpha (fast_color_thres == 0)
        weight = torch.zeros_like(alpha)
        T = torch.ones_like(alpha)
    last = torch.empty(n_rays, dtype=alpha.dtype, device=dev)
    i_start = torch.empty(n_rays, dtype=torch.int64, device=dev)
    i_end = torch.empty(n_rays, dtype=torch.int64, device=dev)
    with _lib.guard(dev):
        _lib.check(_lib.entry("ugrid_alpha2weight", dt)(_p(alpha), _p(ray_id), n, n_rays, _p(weight), _p(T), _p(last), _p(i_start),
                                         _p(i_end), _s(alpha)), "alpha2weight")
    return [weight, T, last, i_start, i_end]


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
    _lib.require_cuda(("alpha", alpha), ("weight", weight), ("T", T), ("alphainv_last", alphainv_last),
                      ("i_start", i_start), ("i_end", i_end), ("grad_weights", grad_weights),
                      ("grad_last", grad_last))
    dt = _lib.real_dtype(("alpha", alpha), ("weight", weight), ("T", T), ("alphainv_last", alphainv_last),
                         ("grad_weights", grad_weights), ("grad_last", grad_last))
    grad = torch.empty_like(alpha)
    with _lib.guard(alpha.device):
        _lib.check(_lib.entry("ugrid_alpha2weight_backward", dt)(_p(alpha), _p(weight), _p(T), _p(alphainv_last), _p(i_start),
                                                  _p(i_end), alpha.size(0), int(n_rays), _p(grad_weights),
                                                  _p(grad_last), _p(grad), _s(alpha)), "alpha2weight_backward")
    return grad
