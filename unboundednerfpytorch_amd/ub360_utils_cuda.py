"""MI355X drop-in for `ub360_utils_cuda` (/root/reference/FourierGrid/cuda/ub360_utils.cpp:21)."""
import torch

from . import _lib

_L = _lib.load()


def cumdist_thres(dist, thres):
    _lib.require_cuda(("dist", dist))
    _lib.require_f32(("dist", dist))
    if dist.dim() != 2:
        raise RuntimeError("dist must be [n_rays, n_pts]")
    mask = torch.empty(dist.size(0), dist.size(1), dtype=torch.bool, device=dist.device)
    with torch.cuda.device(dist.device):
        _lib.check(_L.ugrid_cumdist_thres(_lib.ptr(dist), float(thres), dist.size(0), dist.size(1), _lib.ptr(mask),
                                          _lib.stream_of(dist)), "cumdist_thres")
    return mask
