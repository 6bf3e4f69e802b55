"""MI355X drop-in for `ub360_utils_cuda` (/root/reference/FourierGrid/cuda/ub360_utils.cpp:21), plus the
`segment_cumsum` the reference's DistortionLoss calls but its extension never exported."""
import torch

from . import _lib

_L = _lib.load()


def cumdist_thres(dist, thres):
    _lib.require_cuda(("dist", dist))
    dt = _lib.real_dtype(("dist", dist))
    if dist.dim() != 2:
        raise RuntimeError("dist must be [n_rays, n_pts]")
    mask = torch.empty(dist.size(0), dist.size(1), dtype=torch.bool, device=dist.device)
    with _lib.guard(dist.device):
        _lib.check(_lib.entry("ugrid_cumdist_thres", dt)(_lib.ptr(dist), float(thres), dist.size(0), dist.size(1), _lib.ptr(mask),
                                          _lib.stream_of(dist)), "cumdist_thres")
    return mask


def segment_cumsum(w, s, ray_id, n_rays=None):
    """(w_prefix, w_total, ws_prefix, ws_total) for DistortionLoss (FourierGrid_model.py:684-708): exclusive running
    sums of w and w*s inside each ray of the sorted ray_id, and per-ray totals.  n_rays defaults to
    ray_id[-1] + 1 (one host read, where the reference does `ray_id.max()+1`)."""
    _lib.require_cuda(("w", w), ("s", s), ("ray_id", ray_id))
    _lib.require_f32(("w", w), ("s", s))
    if ray_id.dtype != torch.int64 or w.dim() != 1 or w.shape != s.shape or w.shape != ray_id.shape:
        raise RuntimeError("w, s [n] float32 and ray_id [n] int64 expected")
    n = w.numel()
    if n_rays is None:
        n_rays = int(ray_id[-1].item()) + 1 if n > 0 else 0
    dev = w.device
    w_prefix, ws_prefix = torch.empty_like(w), torch.empty_like(w)
    w_total = torch.zeros(n_rays, dtype=torch.float32, device=dev)
    ws_total = torch.zeros(n_rays, dtype=torch.float32, device=dev)
    if n_rays == 0:
        return w_prefix, w_total, ws_prefix, ws_total
    scratch = torch.empty(2 * n_rays, dtype=torch.int64, device=dev)
    with _lib.guard(dev):
        _lib.check(_L.ugrid_segment_cumsum(_lib.ptr(w), _lib.ptr(s), _lib.ptr(ray_id), n, n_rays, _lib.ptr(w_prefix),
                                           _lib.ptr(w_total), _lib.ptr(ws_prefix), _lib.ptr(ws_total),
                                           _lib.ptr(scratch), _lib.stream_of(w)), "segment_cumsum")
    return w_prefix, w_total, ws_prefix, ws_total
