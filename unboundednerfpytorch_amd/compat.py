"""Register this package's modules under the names the reference imports at module top
(`import render_utils_cuda`, `import total_variation_cuda`, `import ub360_utils_cuda`,
`import adam_upd_cuda` -- dvgo.py:13, grid.py:10-11, FourierGrid_grid.py:10-11, dcvgo.py:15,
FourierGrid_model.py:17-18, masked_adam.py:3), so run_train.py / run_render.py / MaskedAdam of the
reference call the MI355X kernels unchanged.  Call before importing any reference model file."""
import sys


def install_as_reference_extensions():
    from . import adam_upd_cuda, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    for mod in (render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda):
        sys.modules[mod.__name__.rsplit(".", 1)[-1]] = mod
    return ("render_utils_cuda", "total_variation_cuda", "ub360_utils_cuda", "adam_upd_cuda")
