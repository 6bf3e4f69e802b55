"""Register this package's modules under the names the reference imports at module top
(`import render_utils_cuda`, `import total_variation_cuda`, `import ub360_utils_cuda`,
`import adam_upd_cuda` -- dvgo.py:13, grid.py:10-11, FourierGrid_grid.py:10-11, dcvgo.py:15,
FourierGrid_model.py:17-18, masked_adam.py:3), so run_train.py / run_render.py / MaskedAdam of the
reference call the MI355X kernels unchanged.  Call before importing any reference model file."""
import sys


def install_as_reference_extensions(native=False):
    """native=False: the ctypes mirrors of this package (always available).  native=True: the pybind11 modules of binding/
    (what a maintainer would build in place of FourierGrid/cuda/*.cu; `python binding/build.py`)."""
    if native:
        import os
        import sys as _sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if root not in _sys.path:
            _sys.path.insert(0, root)
        from binding import build as _b
        if not _b.available():
            raise RuntimeError("binding/_build is empty: run `python binding/build.py` (needs torch headers and libugrid_hip.so)")
        return tuple(_b.install())
    from . import adam_upd_cuda, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    for mod in (render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda):
        sys.modules[mod.__name__.rsplit(".", 1)[-1]] = mod
    return ("render_utils_cuda", "total_variation_cuda", "ub360_utils_cuda", "adam_upd_cuda")


def install_model_classes(package="FourierGrid"):
    """Rebind the reference's three model classes to this package's training models -- `<package>.dvgo.DirectVoxGO`,
    `<package>.dcvgo.DirectContractedVoxGO`, `<package>.FourierGrid_model.FourierGridModel` -- so that the reference's training
    program builds them (run_train.py:19-56 create_new_model), dispatches on them (`isinstance` at :191-196, :380) and checkpoints
    them unchanged, with the fused training forwards of this package underneath.  Call after install_as_reference_extensions(),
    with the reference tree on sys.path, BEFORE importing `<package>.run_train` (which binds FourierGridModel by name at import,
    run_train.py:10).  Returns the three original classes (to restore them: setattr them back)."""
    import importlib
    from . import fourier_model, voxgo_model
    dv = importlib.import_module(package + ".dvgo")
    dc = importlib.import_module(package + ".dcvgo")
    fg = importlib.import_module(package + ".FourierGrid_model")
    orig = (dv.DirectVoxGO, dc.DirectContractedVoxGO, fg.FourierGridModel)
    dv.DirectVoxGO = voxgo_model.DirectVoxGO
    dc.DirectContractedVoxGO = voxgo_model.DirectContractedVoxGO
    fg.FourierGridModel = fourier_model.FourierGridModel
    rt = sys.modules.get(package + ".run_train")
    if rt is not None and hasattr(rt, "FourierGridModel"):       # already imported: rebind its by-name import too
        rt.FourierGridModel = fourier_model.FourierGridModel
    return orig
