"""oracle/ref_ops.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/_build/liboracle.so (the C restatement in oracle/ref_ops.c, and of the
double instantiation in oracle/ref_ops_f64.c: float64 tensors go there, like the reference's
AT_DISPATCH_FLOATING_TYPES) that exposes, on CPU torch tensors, exactly the 18 functions of the reference's four
pybind modules (signatures: /root/reference/FourierGrid/cuda/render_utils.cpp:170-184,
total_variation.cpp:23, ub360_utils.cpp:21, adam_upd.cpp:79-86).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (unboundednerfpytorch_amd) never does.
"""
import ctypes
import os
import subprocess
import types

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile oracle/ref_ops.c with gcc (recipe: oracle/Makefile)."""
    srcs = [os.path.join(_HERE, "ref_ops.c"), os.path.join(_HERE, "ref_ops_f64.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_sample_pts_on_rays_count.restype = ctypes.c_int64
        _lib.orc64_sample_pts_on_rays_count.restype = ctypes.c_int64
        _lib.orc_adam_step_size.restype = ctypes.c_float
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _f(x):
    # pybind converts 0-d / 1-element tensors through __float__
    return ctypes.c_float(float(x))


def _i64(x):
    return ctypes.c_int64(int(x))


def _chk(*ts):
    for t in ts:
        assert t.device.type == "cpu" and t.is_contiguous(), "oracle takes contiguous CPU tensors"


def _chk32(*ts):
    _chk(*ts)
    for t in ts:
        assert t.dtype == torch.float32, "this op has no reference counterpart: fp32 only"


def _real(name, *ts):
    """the C function restating `name` for the tensors' floating type: orc_<name> (float, ref_ops.c) or orc64_<name> (double,
    ref_ops_f64.c) -- one type per call, like the reference's dispatch on the first tensor"""
    _chk(*ts)
    dt = ts[0].dtype
    assert dt in (torch.float32, torch.float64) and all(t.dtype == dt for t in ts), "one floating type (float32 / float64) per call"
    return getattr(lib(), ("orc64_" if dt == torch.float64 else "orc_") + name)


# ------------------------------------------------------------------ render_utils_cuda
def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    fn = _real("infer_t_minmax", rays_o, rays_d, xyz_min, xyz_max)
    n = rays_o.shape[0]
    t_min = torch.empty(n, dtype=rays_o.dtype)
    t_max = torch.empty(n, dtype=rays_o.dtype)
    fn(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), _f(near), _f(far),
                             _i64(n), _p(t_min), _p(t_max))
    return [t_min, t_max]


def infer_n_samples(rays_d, t_min, t_max, stepdist):
    fn = _real("infer_n_samples", rays_d, t_min, t_max)
    n = t_min.shape[0]
    out = torch.empty(n, dtype=torch.int64)
    fn(_p(rays_d), _p(t_min), _p(t_max), _f(stepdist), _i64(n), _p(out))
    return out


def infer_ray_start_dir(rays_o, rays_d, t_min):
    fn = _real("infer_ray_start_dir", rays_o, rays_d, t_min)
    n = rays_o.shape[0]
    start = torch.empty_like(rays_o)
    dirs = torch.empty_like(rays_o)
    fn(_p(rays_o), _p(rays_d), _p(t_min), _i64(n), _p(start), _p(dirs))
    return [start, dirs]


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    fn = _real("sample_pts_on_rays_count", rays_o, rays_d, xyz_min, xyz_max)
    fill = _real("sample_pts_on_rays_fill", rays_o)
    n = rays_o.shape[0]
    t_min = torch.empty(n, dtype=rays_o.dtype)
    t_max = torch.empty(n, dtype=rays_o.dtype)
    n_steps = torch.empty(n, dtype=torch.int64)
    total = fn(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max),
                                               _f(near), _f(far), _f(stepdist), _i64(n),
                                               _p(t_min), _p(t_max), _p(n_steps))
    pts = torch.empty(total, 3, dtype=rays_o.dtype)
    mask = torch.empty(total, dtype=torch.bool)
    ray_id = torch.empty(total, dtype=torch.int64)
    step_id = torch.empty(total, dtype=torch.int64)
    fill(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), _p(t_min),
                                      _p(n_steps), _f(stepdist), _i64(n), _i64(total),
                                      _p(pts), _p(mask), _p(ray_id), _p(step_id))
    return [pts, mask, ray_id, step_id, n_steps, t_min, t_max]


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    fn = _real("sample_ndc_pts_on_rays", rays_o, rays_d, xyz_min, xyz_max)
    n = rays_o.shape[0]
    pts = torch.empty(n, N_samples, 3, dtype=rays_o.dtype)
    mask = torch.empty(n, N_samples, dtype=torch.bool)
    fn(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max),
                                     _i64(N_samples), _i64(n), _p(pts), _p(mask))
    return [pts, mask]


def sample_bg_pts_on_rays(rays_o, rays_d, t_max, bg_preserve, N_samples):
    fn = _real("sample_bg_pts_on_rays", rays_o, rays_d, t_max)
    n = rays_o.shape[0]
    pts = torch.empty(n, N_samples, 3, dtype=rays_o.dtype)
    fn(_p(rays_o), _p(rays_d), _p(t_max), _f(bg_preserve),
                                    _i64(N_samples), _i64(n), _p(pts))
    return pts


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    _chk(world)
    fn = _real("maskcache_lookup", xyz, xyz2ijk_scale, xyz2ijk_shift)
    assert world.dtype == torch.bool
    n = xyz.shape[0]
    out = torch.zeros(n, dtype=torch.bool)
    fn(_p(world), _p(xyz), _p(xyz2ijk_scale), _p(xyz2ijk_shift),
                               _i64(world.shape[0]), _i64(world.shape[1]), _i64(world.shape[2]),
                               _i64(n), _p(out))
    return out


def raw2alpha(density, shift, interval):
    fn = _real("raw2alpha", density)
    n = density.shape[0]
    exp_d = torch.empty_like(density)
    alpha = torch.empty_like(density)
    fn(_p(density), _f(shift), _f(interval), _p(None), _i64(n), _p(exp_d), _p(alpha))
    return [exp_d, alpha]


def raw2alpha_nonuni(density, shift, interval):
    fn = _real("raw2alpha", density, interval)
    n = density.shape[0]
    exp_d = torch.empty_like(density)
    alpha = torch.empty_like(density)
    fn(_p(density), _f(shift), _f(0.0), _p(interval), _i64(n), _p(exp_d), _p(alpha))
    return [exp_d, alpha]


def raw2alpha_backward(exp_d, grad_back, interval):
    fn = _real("raw2alpha_backward", exp_d, grad_back)
    grad = torch.empty_like(exp_d)
    fn(_p(exp_d), _p(grad_back), _f(interval), _p(None),
                                 _i64(exp_d.shape[0]), _p(grad))
    return grad


def raw2alpha_nonuni_backward(exp_d, grad_back, interval):
    fn = _real("raw2alpha_backward", exp_d, grad_back, interval)
    grad = torch.empty_like(exp_d)
    fn(_p(exp_d), _p(grad_back), _f(0.0), _p(interval),
                                 _i64(exp_d.shape[0]), _p(grad))
    return grad


def alpha2weight(alpha, ray_id, n_rays):
    fn = _real("alpha2weight", alpha)
    _chk(ray_id)
    assert ray_id.dtype == torch.int64
    n = alpha.shape[0]
    weight = torch.empty_like(alpha)
    T = torch.empty_like(alpha)
    last = torch.empty(n_rays, dtype=alpha.dtype)
    i_start = torch.empty(n_rays, dtype=torch.int64)
    i_end = torch.empty(n_rays, dtype=torch.int64)
    fn(_p(alpha), _p(ray_id), _i64(n), _i64(n_rays), _p(weight), _p(T), _p(last),
                           _p(i_start), _p(i_end))
    return [weight, T, last, i_start, i_end]


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
    grad_weights = grad_weights.contiguous()
    grad_last = grad_last.contiguous()
    fn = _real("alpha2weight_backward", alpha, weight, T, alphainv_last, grad_weights, grad_last)
    grad = torch.empty_like(alpha)
    fn(_p(alpha), _p(weight), _p(T), _p(alphainv_last), _p(i_start), _p(i_end),
                                    _i64(alpha.shape[0]), _i64(n_rays), _p(grad_weights), _p(grad_last),
                                    _p(grad))
    return grad


# ------------------------------------------------------------------ total_variation_cuda
def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    fn = _real("total_variation_add_grad", param, grad)
    assert param.dim() == 5 and param.shape == grad.shape
    fn(_p(param), _p(grad), _f(wx), _f(wy), _f(wz),
                                       ctypes.c_int(1 if dense_mode else 0),
                                       _i64(param.shape[2]), _i64(param.shape[3]), _i64(param.shape[4]),
                                       _i64(param.numel()))


# ------------------------------------------------------------------ ub360_utils_cuda
def segment_cumsum(w, s, ray_id, n_rays=None):
    """(w_prefix, w_total, ws_prefix, ws_total): see orc_segment_cumsum (the reference calls this op but does not
    ship it; FourierGrid_model.py:684-708)."""
    _chk32(w, s)
    assert ray_id.dtype == torch.int64 and w.shape == s.shape == ray_id.shape and w.dim() == 1
    n = w.shape[0]
    if n_rays is None:
        n_rays = int(ray_id.max()) + 1 if n > 0 else 0
    w_prefix, ws_prefix = torch.empty_like(w), torch.empty_like(w)
    w_total, ws_total = torch.zeros(n_rays, dtype=torch.float32), torch.zeros(n_rays, dtype=torch.float32)
    lib().orc_segment_cumsum(_p(w), _p(s), _p(ray_id), _i64(n), _i64(n_rays), _p(w_prefix), _p(w_total),
                             _p(ws_prefix), _p(ws_total))
    return w_prefix, w_total, ws_prefix, ws_total


def cumdist_thres(dist, thres):
    fn = _real("cumdist_thres", dist)
    mask = torch.zeros(dist.shape[0], dist.shape[1], dtype=torch.bool)
    fn(_p(dist), _f(thres), _i64(dist.shape[0]), _i64(dist.shape[1]), _p(mask))
    return mask


# ------------------------------------------------------------------ adam_upd_cuda
def _adam(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps, mode):
    ts = [param, grad, exp_avg, exp_avg_sq] + ([perlr] if perlr is not None else [])
    fn = _real("adam_upd", *ts)
    fn(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(perlr), _i64(param.numel()),
                       ctypes.c_int(int(step)), _f(beta1), _f(beta2), _f(lr), _f(eps), ctypes.c_int(mode))


def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    _adam(param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps, 0)


def masked_adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    _adam(param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps, 1)


def adam_upd_with_perlr(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    _adam(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps, 2)


def adam_step_size(step, beta1, beta2, lr):
    return float(lib().orc_adam_step_size(ctypes.c_int(int(step)), _f(beta1), _f(beta2), _f(lr)))


# ------------------------------------------------------------------ module-shaped views
def _mod(name, fns):
    m = types.ModuleType(name)
    m.__doc__ = "CPU oracle stand-in for the reference extension module %r (test infrastructure)" % name
    for f in fns:
        setattr(m, f.__name__, f)
    return m


render_utils_cuda = _mod("render_utils_cuda", [
    infer_t_minmax, infer_n_samples, infer_ray_start_dir, sample_pts_on_rays, sample_ndc_pts_on_rays,
    sample_bg_pts_on_rays, maskcache_lookup, raw2alpha, raw2alpha_nonuni, raw2alpha_backward,
    raw2alpha_nonuni_backward, alpha2weight, alpha2weight_backward])
total_variation_cuda = _mod("total_variation_cuda", [total_variation_add_grad])
ub360_utils_cuda = _mod("ub360_utils_cuda", [cumdist_thres, segment_cumsum])
adam_upd_cuda = _mod("adam_upd_cuda", [adam_upd, masked_adam_upd, adam_upd_with_perlr])
