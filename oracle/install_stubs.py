"""oracle/install_stubs.py -- TEST INFRASTRUCTURE ONLY.

Makes the reference's own Python model files importable and runnable on CPU *in the build
container* (where /root/reference exists) by registering stand-ins in sys.modules for the
modules they import unconditionally at the top:

  render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda
      -> the C oracle (oracle/ref_ops.py)   [dvgo.py:13, grid.py:10-11, masked_adam.py:3 ...]
  torch_scatter.segment_coo(src, index, out, reduce='sum')
      -> out.index_add_(0, index, src)       [FourierGrid_model.py:640; third-party, un-vendored]
  cv2, imageio, mmengine, lpips ...          -> empty placeholders (never called on this path)

Used only by tests/golden/gen_golden.py (fixture generation) and by tests that are skipped
when /root/reference is absent (i.e. on the GPU box).
"""
import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("UNERF_REFERENCE_ROOT", "/root/reference")
OPS_BACKEND = None   # default back-end of install(): None = the CPU oracle (tests/test_gpu_reference_callers.py sets the HIP modules)


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "FourierGrid"))


def _segment_coo(src, index, out=None, dim_size=None, reduce="sum"):
    assert reduce == "sum"
    if out is None:
        shape = list(src.shape)
        shape[0] = int(dim_size)
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def _scatter_add(src, index, dim=0, out=None, dim_size=None):
    if out is None:
        shape = list(src.shape)
        shape[dim] = int(dim_size)
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


def install(ops_backend=None):
    """Register the stubs.  `ops_backend` may provide the four extension modules
    (an object with attributes render_utils_cuda, total_variation_cuda, ub360_utils_cuda,
    adam_upd_cuda); default is the CPU oracle."""
    if ops_backend is None:
        ops_backend = OPS_BACKEND
    if ops_backend is None:
        from oracle import ref_ops as ops_backend
    for name in ("render_utils_cuda", "total_variation_cuda", "ub360_utils_cuda", "adam_upd_cuda"):
        sys.modules[name] = getattr(ops_backend, name)
    ts = types.ModuleType("torch_scatter")
    ts.segment_coo = _segment_coo
    ts.scatter_add = _scatter_add      # imported (never called on this path) by dmpigo.py:11
    sys.modules["torch_scatter"] = ts
    for name in ("cv2", "imageio", "lpips", "mmengine", "mmcv", "torch_efficient_distloss"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if reference_available() and REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def import_reference(modname):
    """import FourierGrid.<modname> from the reference tree (build container only)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install()
    return importlib.import_module("FourierGrid." + modname)
