"""oracle/ref_model.py -- TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg).

The reference's OWN `FourierGridModel` (FourierGrid/FourierGrid_model.py:29-672), instantiated from the plain `state`
dict the renderer takes and run through its own `forward` (:554-672), over one of two back-ends for the four extension
modules it imports (render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda):

  backend = "kernels:fma" | "kernels:nofma"   the reference's own FourierGrid/cuda/*.cu compiled for gfx950 by
                                              oracle/build_ref.py (oracle/_ref/<variant>/*.so) -- needs a GPU.  This is
                                              "the reference executing on the MI355X": its Python, its kernels,
                                              torch-ROCm's grid_sample / Linear.
  backend = "oracle"                          the C restatement (oracle/ref_ops.py) -- the reference has no CPU path of
                                              its own; this is BASELINE.md section 3's CPU baseline.

The reference's Python files come from /root/reference when present (build container) or from the archive
oracle/build_ref.py staged under oracle/_ref/reference_py.tar (git-ignored, shipped to the GPU box by gpurun).
Nothing here is imported by the product.
"""
import importlib
import sys
import types

import torch

_REF_ROOT = None
_LOADED_BACKEND = None


def available(backend="oracle"):
    """Can `reference_model(..., backend)` run here?  (reference Python present; for "kernels:*" also the built .so)"""
    from oracle import build_ref
    import os
    root = os.environ.get("UNERF_REFERENCE_ROOT", "/root/reference")
    have_py = os.path.isdir(os.path.join(root, "FourierGrid")) or os.path.exists(os.path.join(build_ref.OUT, "reference_py.tar"))
    if not have_py:
        return False
    if backend.startswith("kernels:"):
        return build_ref.built(backend.split(":", 1)[1]) and torch.cuda.is_available()
    return True


def _import_reference(backend):
    """import FourierGrid.FourierGrid_model with `backend`'s extension modules registered under the reference's names."""
    global _REF_ROOT, _LOADED_BACKEND
    from oracle import build_ref, install_stubs
    if _REF_ROOT is None:
        _REF_ROOT = build_ref.reference_python_root()
        if _REF_ROOT is None:
            raise RuntimeError("reference Python not available (no /root/reference, no oracle/_ref/reference_py.tar)")
    if backend.startswith("kernels:"):
        mods = build_ref.load(backend.split(":", 1)[1])
        ops = types.SimpleNamespace(**mods)
    elif backend == "oracle":
        from oracle import ref_ops as ops
    else:
        raise ValueError(backend)
    if _LOADED_BACKEND != backend:
        # the reference binds the extension modules at import time (dvgo.py:13, grid.py:10-11, ...): re-import per backend
        for m in [k for k in sys.modules if k == "FourierGrid" or k.startswith("FourierGrid.")]:
            del sys.modules[m]
    old = install_stubs.REFERENCE_ROOT
    install_stubs.REFERENCE_ROOT = _REF_ROOT
    try:
        install_stubs.install(ops)
        mod = importlib.import_module("FourierGrid.FourierGrid_model")
    finally:
        install_stubs.REFERENCE_ROOT = old
    _LOADED_BACKEND = backend
    return mod


def reference_model(state, device, backend):
    """The reference's FourierGridModel holding `state`'s parameters, in eval mode on `device`."""
    mod = _import_reference(backend)
    G = int(state["world_len"])
    C = int(state["k0_grid"].shape[1]) if len(state["rgbnet_weights"]) else 0
    bg = float(state["bg_len"])
    # the model is built on the CPU with 1-voxel grids swapped in afterwards: the constructor would allocate (and
    # zero-fill) full-size grids that are thrown away again
    model = mod.FourierGridModel(
        xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1],
        num_voxels_density=G ** 3, num_voxels_base_density=G ** 3, num_voxels_rgb=G ** 3, num_voxels_base_rgb=G ** 3,
        num_voxels_viewdir=-1, alpha_init=1e-4, fast_color_thres=float(state["fast_color_thres"]),
        contracted_norm=state.get("contracted_norm", "inf"), bg_len=bg,
        fourier_freq_num=int(state["fourier_freq_num"]), rgbnet_dim=C, viewbase_pe=int(state["viewbase_pe"]))
    assert int(model.world_len_density) == G, (int(model.world_len_density), G)
    assert abs(float(model.act_shift) - float(state["act_shift"])) < 1e-6
    assert abs(float(model.voxel_size_ratio_density) - float(state["voxel_size_ratio"])) < 1e-6
    dev = torch.device(device)
    with torch.no_grad():
        model.density.grid = torch.nn.Parameter(state["density_grid"].to(dev, torch.float32), requires_grad=False)
        model.k0.grid = torch.nn.Parameter(state["k0_grid"].to(dev, torch.float32), requires_grad=False)
    model = model.to(dev).eval()
    if C:
        lin = [m for m in model.rgbnet.modules() if isinstance(m, torch.nn.Linear)]
        assert len(lin) == len(state["rgbnet_weights"])
        with torch.no_grad():
            for m, w, b in zip(lin, state["rgbnet_weights"], state["rgbnet_biases"]):
                m.weight.copy_(w.to(dev))
                m.bias.copy_(b.to(dev))
    return model


@torch.no_grad()
def render(model, rays_o, rays_d, viewdirs, stepsize, chunk=8192):
    """run_render.py:52-58: the reference's own chunked render loop (8192 rays per forward), per-ray outputs only."""
    keys = ("rgb_marched", "depth", "alphainv_last")
    outs = {k: [] for k in keys}
    # the reference creates its sample table without a device (FourierGrid_model.py:526-532) and relies on the program
    # having made CUDA the default tensor type (run_FourierGrid.py: torch.set_default_tensor_type('torch.cuda.FloatTensor'))
    with torch.device(rays_o.device):
        for b in range(0, rays_o.shape[0], chunk):
            r = model(rays_o[b:b + chunk], rays_d[b:b + chunk], viewdirs[b:b + chunk], stepsize=stepsize, render_depth=True)
            for k in keys:
                outs[k].append(r[k])
    return {k: torch.cat(v) for k, v in outs.items()}
