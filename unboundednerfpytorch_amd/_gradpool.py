"""Zero-initialised gradient buffers of the grid parameters, recycled between training steps.

The backward of a grid lookup scatters ~1e5 samples into a gradient array the size of the grid (3.46 GB for the S3 k0 grid):
filling that array with zeros is a full HBM pass (0.55 ms) per step.  The fused dense TV + Adam pass reads every gradient
element anyway and can write the zeros back where it found something else (`rezero_grad`, include/ugrid_hip.h) -- a few per
cent of the lines.  The optimizer then parks the buffer here (`give`) and the next backward of the SAME parameter picks it
up (`take`) instead of allocating and filling a new one.

Invariants: a parked buffer is all zero, has the parameter's shape, strides, dtype and device, and nothing else references
it (AccumulateGrad adopts a gradient without copying only while it holds the sole reference).  One buffer per parameter;
entries die with their parameter.

Touched-line bitmaps (channel-last grids): the backward also marks which 256-byte lines of the buffer it adds to
(`touch_for_backward`, include/ugrid_hip.h: ugrid_grid_query_backward_cl_touch), and the optimizer's masked TV / masked Adam /
fused dense pass visit only those (`touch_of`).  Invariant: an UNSET bit means the line is all zero.  Stale SET bits are
harmless (the line is read and found zero), so the bitmap is only ever cleared by a pass that has just re-zeroed the buffer.
The bitmap is bound to the buffer's address and to its most recent backward: a gradient that autograd accumulated from two
backward calls, or any tensor that is not the very buffer the last backward filled, gets no bitmap and the scanning kernels.

The address alone does not prove that: autograd's AccumulateGrad sums a second producer's contribution IN PLACE into the
first-arrived tensor, so `p.pow(2).sum() + GridQuery(p)` leaves `.grad` at the lookup's buffer address with dense values the
bitmap knows nothing about.  A bitmap is therefore only created, marked and served for parameters the training step has
CERTIFIED for the current backward (`certify`: "this parameter's only gradient producer in this graph is one marking lookup") --
`train_step.train_iteration` does that for the model's own loss graph -- and only when exactly one marking backward has run since.
Everything else (the drop-in MaskedAdam on a user's own graph, world > 1, injected ops) gets the scanning kernels."""
import weakref

import torch

_POOL = {}          # id(param) -> (weakref to param, buffer)
_TOUCH = {}         # id(param) -> [data_ptr of the buffer, bitmap (int32 tensor), numel]
_CERT = {}          # id(param) -> marking backward calls since certify() (the bitmap is valid only at exactly 1)
_FINAL = set()      # ids of parameters that already carry the clean-up finalizer (ONE per parameter, not one per training step)
enabled = True
touch_enabled = True


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


def certify(params):
    """The caller vouches, for the backward it is about to run, that each of `params` receives gradient from exactly ONE
    marking grid lookup and from nothing else (no regulariser on the raw grid, no second lookup, no hook that edits .grad).
    Only certified parameters get touched-line bitmaps.  Pair with `decertify` (try / finally) once the optimizer has stepped."""
    for p in params:
        if isinstance(p, torch.nn.Parameter) and p.requires_grad:
            k = id(p)
            if k not in _FINAL:
                # decertify() pops the _CERT entry after every training step, so "k not in _CERT" held at every step: two new
                # finalizers per certified grid per iteration, ~80 k objects over a 40 k-step run, all firing at pg_scale or exit
                # (ADVICE r4).  One finalizer per parameter object, registered once, cleans all three tables.
                _FINAL.add(k)
                weakref.finalize(p, _forget, k)
            _CERT[k] = 0


def _forget(k):
    _CERT.pop(k, None)
    _TOUCH.pop(k, None)
    _FINAL.discard(k)


def decertify(params):
    for p in params:
        _CERT.pop(id(p), None)


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


def touch_for_backward(key, buf, lib):
    """the bitmap the backward of parameter `key` must mark while it scatters into `buf` (all zero on entry): the buffer's
    own bitmap if it has one, else a fresh cleared one.  None when bitmaps are off or `key` is not poolable."""
    if key is None or not (enabled and touch_enabled) or key not in _CERT:
        _TOUCH.pop(key, None)      # this backward marks nothing: a bitmap of the buffer would no longer describe it
        if key in _CERT:
            _CERT[key] += 2        # ... and an unmarked contribution voids the certificate for this backward
        return None
    _CERT[key] += 1
    ent = _TOUCH.get(key)
    if ent is not None and ent[0] == buf.data_ptr() and ent[2] == buf.numel() and ent[1].device == buf.device:
        return ent[1]
    words = int(lib.ugrid_touch_words(buf.numel()))
    t = torch.zeros(words, dtype=torch.int32, device=buf.device)
    _TOUCH[key] = [buf.data_ptr(), t, buf.numel()]
    return t


def touch_of(param, g):
    """the valid bitmap of gradient `g` of `param` (see the module docstring), or None"""
    if not (enabled and touch_enabled) or g is None:
        return None
    if _CERT.get(id(param)) != 1:      # not certified for this backward, or more than one lookup contributed
        return None
    ent = _TOUCH.get(id(param))
    if ent is None or g is not param.grad or ent[0] != g.data_ptr() or ent[2] != g.numel() or g.stride() != param.stride():
        return None
    return ent[1]


def consume_touch(param):
    """the bitmap of `param`'s gradient has been read AND cleared by a consumer that did not re-zero the buffer (the data-parallel
    exchange, sharded_adam._line_bits): it no longer describes the gradient, so nobody else may be served it in this step"""
    _TOUCH.pop(id(param), None)


def clear():
    _TOUCH.clear()
    _CERT.clear()
    for k, (ref, _) in list(_POOL.items()):
        _POOL[k] = (ref, None)
