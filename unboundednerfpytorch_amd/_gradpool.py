"""Zero-initialised gradient buffers of the grid parameters, recycled between training steps.

The backward of a grid lookup scatters ~1e5 samples into a gradient array the size of the grid (3.46 GB for the S3 k0 grid):
filling that array with zeros is a full HBM pass (0.55 ms) per step.  The fused dense TV + Adam pass reads every gradient
element anyway and can write the zeros back where it found something else (`rezero_grad`, include/ugrid_hip.h) -- a few per
cent of the lines.  The optimizer then parks the buffer here (`give`) and the next backward of the SAME parameter picks it
up (`take`) instead of allocating and filling a new one.

Invariants: a parked buffer is all zero, has the parameter's shape, strides, dtype and device, and nothing else references
it (AccumulateGrad adopts a gradient without copying only while it holds the sole reference).  One buffer per parameter;
entries die with their parameter."""
import weakref

import torch

_POOL = {}          # id(param) -> (weakref to param, buffer)
enabled = True


def key_of(t):
    """pool key of a grid parameter at forward time (None for anything that is not a leaf parameter on a GPU)"""
    if enabled and isinstance(t, torch.nn.Parameter) and t.is_cuda and t.requires_grad:
        return id(t)
    return None


def give(param, buf):
    """park an all-zero gradient buffer of `param` (the caller has dropped every other reference, param.grad included)"""
    if not enabled or buf.shape != param.shape or buf.stride() != param.stride() or buf.device != param.device \
            or buf.dtype != param.dtype:
        return False
    k = id(param)
    if k not in _POOL:
        weakref.finalize(param, _POOL.pop, k, None)
    _POOL[k] = (weakref.ref(param), buf)
    return True


def take(key, shape, stride, device):
    """the parked buffer for `key` if it still fits, else None (the caller allocates zeros)"""
    ent = _POOL.get(key) if key is not None else None
    if ent is None or ent[1] is None:
        return None
    ref, buf = ent
    if ref() is None or tuple(buf.shape) != tuple(shape) or buf.stride() != tuple(stride) or buf.device != device:
        _POOL[key] = (ref, None)
        return None
    _POOL[key] = (ref, None)
    return buf


def clear():
    for k, (ref, _) in list(_POOL.items()):
        _POOL[k] = (ref, None)
