"""MI355X drop-in for `total_variation_cuda` (/root/reference/FourierGrid/cuda/total_variation.cpp:23)."""
import torch

from . import _lib

_L = _lib.load()


def total_variation_add_grad_touched(param, grad, wx, wy, wz, touch):
    """NEW (not in the reference module): total_variation_add_grad(..., dense_mode=False) on channel-last storage, visiting
    only the 256-byte lines of `grad` that its touched-line bitmap `touch` marks (_gradpool.touch_of; the lookup backward
    sets the bits).  Same results as the scanning call."""
    return _tv(param, grad, wx, wy, wz, False, touch)


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    """grad += TV gradient of param, in place; sizes from param.size(2..4) like the reference
    (total_variation_kernel.cu:39-41), so [P,C,X,Y,Z] grids work.  Returns None."""
    return _tv(param, grad, wx, wy, wz, dense_mode, None)


def _tv(param, grad, wx, wy, wz, dense_mode, touch):
    cl = _lib.require_cuda_grid(("param", param), ("grad", grad))
    dt = _lib.real_dtype(("param", param), ("grad", grad))
    if param.dim() != 5 or param.shape != grad.shape:
        raise RuntimeError("param/grad must be 5-D tensors of equal shape")
    if dt == torch.float64:      # the reference's double instantiation (total_variation_kernel.cu:50,59): canonical layout only
        if cl or touch is not None:
            raise RuntimeError("float64 total_variation_add_grad: canonical (contiguous) layout, no touched-line bitmap")
        with _lib.guard(param.device):
            _lib.check(_lib.entry("ugrid_total_variation_add_grad", dt)(
                _lib.ptr(param), _lib.ptr(grad), float(wx), float(wy), float(wz), 1 if dense_mode else 0, param.size(2), param.size(3),
                param.size(4), param.numel(), _lib.stream_of(param)), "total_variation_add_grad (float64)")
        return
    if touch is not None and (not cl or dense_mode):
        raise RuntimeError("the touched-line bitmap serves the masked mode on channel-last storage only")
    if cl and touch is not None:
        with _lib.guard(param.device):
            rc = _L.ugrid_total_variation_add_grad_cl_touch(_lib.ptr(param), _lib.ptr(grad), float(wx), float(wy), float(wz),
                                                            param.size(2), param.size(3), param.size(4), param.size(1),
                                                            param.numel(), _lib.ptr(touch), _lib.stream_of(param))
        _lib.check(rc, "total_variation_add_grad (touch)")
        return
    if cl:      # channel-last storage [P][X][Y][Z][C] of the same logical tensor (training layout, grid.FourierGrid)
        with _lib.guard(param.device):
            rc = _L.ugrid_total_variation_add_grad_cl(_lib.ptr(param), _lib.ptr(grad), float(wx), float(wy), float(wz),
                                                      1 if dense_mode else 0, param.size(2), param.size(3), param.size(4),
                                                      param.size(1), param.numel(), _lib.stream_of(param))
        if rc == 801:
            raise RuntimeError("channel-last total_variation_add_grad needs C % 4 == 0 and fewer than 2^31 elements")
        _lib.check(rc, "total_variation_add_grad")
        return
    with _lib.guard(param.device):
        _lib.check(_L.ugrid_total_variation_add_grad(_lib.ptr(param), _lib.ptr(grad), float(wx), float(wy), float(wz),
                                                     1 if dense_mode else 0, param.size(2), param.size(3),
                                                     param.size(4), param.numel(), _lib.stream_of(param)),
                   "total_variation_add_grad")
