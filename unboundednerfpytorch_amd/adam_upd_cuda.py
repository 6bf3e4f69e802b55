"""MI355X drop-in for `adam_upd_cuda` (/root/reference/FourierGrid/cuda/adam_upd.cpp:79-86).
All three functions mutate param / exp_avg / exp_avg_sq in place and return None."""
import torch

from . import _lib

_L = _lib.load()


def _run(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps, mode, what):
    named = [("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)]
    if perlr is not None:
        named.append(("perlr", perlr))
    # elementwise over the storage: any dense layout works as long as all operands share it (canonical, or the
    # channel-last training layout of grid.FourierGrid)
    _lib.require_cuda_grid(*named) if param.dim() == 5 else _lib.require_cuda(*named)
    dt = _lib.real_dtype(*named)
    if dt == torch.float64 and mode == 3:
        raise RuntimeError("masked_adam_upd_rezero: float32 only (no reference counterpart)")
    with _lib.guard(param.device):
        _lib.check(_lib.entry("ugrid_adam_upd", dt)(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                     _lib.ptr(perlr), param.numel(), int(step), float(beta1), float(beta2),
                                     float(lr), float(eps), mode, _lib.stream_of(param)), what)


def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    _run(param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps, 0, "adam_upd")


def masked_adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    _run(param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps, 1, "masked_adam_upd")


def masked_adam_upd_rezero(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps, touch=None):
    """NEW (not in the reference module): masked_adam_upd, and `grad` comes back all zero -- its nonzero elements are
    overwritten after use, so the buffer can serve as the next backward's zero-initialised gradient (_gradpool.py).
    touch: the touched-line bitmap of `grad` (_gradpool.touch_of) -- only the marked lines are visited, and the bitmap comes
    back cleared."""
    if touch is None:
        _run(param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps, 3, "masked_adam_upd_rezero")
        return
    named = [("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)]
    _lib.require_cuda_grid(*named) if param.dim() == 5 else _lib.require_cuda(*named)
    _lib.require_f32(*named)
    with _lib.guard(param.device):
        _lib.check(_L.ugrid_masked_adam_upd_touch(_lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                                  param.numel(), int(step), float(beta1), float(beta2), float(lr), float(eps),
                                                  _lib.ptr(touch), _lib.stream_of(param)), "masked_adam_upd_touch")


def adam_upd_multi(items, beta1, beta2, eps, masked):
    """NEW (not in the reference module): adam_upd (masked=False) or masked_adam_upd (masked=True) of several SMALL tensors in ONE
    launch -- items = [(param, grad, exp_avg, exp_avg_sq, step, lr), ...], all contiguous fp32 on one device.  Bit-identical to the
    per-tensor calls (include/ugrid_hip.h: ugrid_adam_upd_multi)."""
    if not items:
        return
    arr = (_lib.AdamItem * len(items))()
    for a, (param, grad, exp_avg, exp_avg_sq, step, lr) in zip(arr, items):
        named = (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq))
        _lib.require_cuda(*named)
        _lib.require_f32(*named)
        a.param, a.grad, a.exp_avg, a.exp_avg_sq = param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr()
        a.numel, a.step, a.lr = param.numel(), int(step), float(lr)
    p0 = items[0][0]
    with _lib.guard(p0.device):
        _lib.check(_L.ugrid_adam_upd_multi(arr, len(items), float(beta1), float(beta2), float(eps), 1 if masked else 0, _lib.stream_of(p0)),
                   "adam_upd_multi")


def adam_upd_with_perlr(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    _run(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps, 2, "adam_upd_with_perlr")


def tv_adam_dense(param, param_out, grad, exp_avg, exp_avg_sq, wx, wy, wz, step, beta1, beta2, lr, eps, skip_zero_grad,
                  rezero_grad=False, touch=None):
    """NEW (not in the reference module): dense total_variation_add_grad + (masked_)adam_upd of one grid parameter
    [.., X, Y, Z] in a single pass (include/ugrid_hip.h: ugrid_tv_adam_dense).  The updated parameter lands in
    `param_out`; `grad` is left untouched, or -- rezero_grad=True -- comes back all zero (its nonzero elements are
    overwritten after use, so the buffer can serve as the next backward's zero-initialised gradient).  Returns False when the shape cannot take the fused path (the caller then
    runs the two reference calls), True otherwise."""
    named = [("param", param), ("param_out", param_out), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)]
    cl = _lib.require_cuda_grid(*named) if param.dim() == 5 else (_lib.require_cuda(*named) or False)
    _lib.require_f32(*named)
    sz_i, sz_j, sz_k = param.shape[-3:]
    flags = int(bool(skip_zero_grad)) | (2 if rezero_grad else 0)
    if cl and touch is not None:       # gradient lines the backward did not mark are known zeros: not read
        with _lib.guard(param.device):
            rc = _L.ugrid_tv_adam_dense_cl_touch(_lib.ptr(param), _lib.ptr(param_out), _lib.ptr(grad), _lib.ptr(exp_avg),
                                                 _lib.ptr(exp_avg_sq), float(wx), float(wy), float(wz), sz_i, sz_j, sz_k, param.shape[1],
                                                 param.numel(), int(step), float(beta1), float(beta2), float(lr), float(eps), flags,
                                                 _lib.ptr(touch), _lib.stream_of(param))
        if rc == 801:
            return False
        _lib.check(rc, "tv_adam_dense (touch)")
        return True
    if cl:
        with _lib.guard(param.device):
            rc = _L.ugrid_tv_adam_dense_cl(_lib.ptr(param), _lib.ptr(param_out), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                           float(wx), float(wy), float(wz), sz_i, sz_j, sz_k, param.shape[1], param.numel(), int(step),
                                           float(beta1), float(beta2), float(lr), float(eps), flags, _lib.stream_of(param))
        if rc == 801:
            return False
        _lib.check(rc, "tv_adam_dense")
        return True
    with _lib.guard(param.device):
        rc = _L.ugrid_tv_adam_dense(_lib.ptr(param), _lib.ptr(param_out), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                    float(wx), float(wy), float(wz), sz_i, sz_j, sz_k, param.numel(), int(step), float(beta1),
                                    float(beta2), float(lr), float(eps), flags, _lib.stream_of(param))
    if rc == 801:      # hipErrorNotSupported
        return False
    _lib.check(rc, "tv_adam_dense")
    return True
