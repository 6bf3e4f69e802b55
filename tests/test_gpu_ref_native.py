"""The HIP library against the REFERENCE'S OWN native kernels.

tests/golden/native_ops.npz holds outputs of FourierGrid/cuda/*.cu (built for gfx950 by oracle/build_ref.py as test
infrastructure, run by tests/golden/gen_native_golden.py) for all 18 functions the four reference modules export.
The drop-in modules of this package, called with the same arguments through the C ABI, must reproduce them BIT FOR
BIT -- including the raw2alpha family, whose expf / powf both sides take from the device libm.  When the compiled
reference (oracle/_ref/, git-ignored, shipped to the GPU box by gpurun) is present, a second test compares live at
8x the sizes, against both its -ffp-contract=off and its default-contraction build."""
import os

import numpy as np
import pytest
import torch

import native_cases as nc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hip_modules():
    from unboundednerfpytorch_amd import adam_upd_cuda, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    return {nc.RU: render_utils_cuda, nc.TV: total_variation_cuda, nc.UB: ub360_utils_cuda, nc.AD: adam_upd_cuda}


def assert_same(want, got, what):
    for name in want:
        assert len(want[name]) == len(got[name]) > 0, name
        for k, (a, b) in enumerate(zip(want[name], got[name])):
            key = "%s %s[%d]" % (what, name, k)
            assert a.shape == b.shape and a.dtype == b.dtype, key
            assert np.array_equal(a.numpy(), b.numpy(), equal_nan=True), key


def test_hip_ops_reproduce_reference_kernel_outputs_bit_for_bit(golden_dir):
    gold, _ = nc.load_golden(os.path.join(golden_dir, "native_ops.npz"))
    got = nc.run_all(hip_modules(), scale=1, device="cuda", chain_from=gold)
    assert_same(gold, got, "golden")


@pytest.mark.parametrize("variant", ["nofma", "fma"])
def test_hip_ops_vs_live_reference_kernels_at_scale(variant):
    from oracle import build_ref
    if not build_ref.built(variant):
        pytest.skip("oracle/_ref/%s not built (python oracle/build_ref.py in the build container)" % variant)
    ref = nc.run_all(build_ref.load(variant), scale=8, device="cuda")
    got = nc.run_all(hip_modules(), scale=8, device="cuda", chain_from=ref)
    assert_same(ref, got, "live " + variant)
    assert ref["sample_pts_on_rays"][0].shape[0] > 15000 and ref["alpha2weight"][0].numel() == 8 * nc.A2W_N


def test_loading_the_reference_kernels_never_overwrites_the_drop_in_modules():
    """The checker must not be able to replace the thing it checks.  CPython re-populates whatever module sits in sys.modules
    under an extension module's NAME when a cached single-phase-init .so of that name is loaded again (import.c:
    import_find_extension); the product registers its drop-in modules under exactly the reference's names
    (compat.install_as_reference_extensions) and oracle/_ref holds same-named pybind modules.  oracle/build_ref.load therefore
    loads every variant once and sets sys.modules' entries aside while it runs: after any number of loads the drop-in modules
    are still this package's Python functions."""
    import sys
    import types
    from oracle import build_ref
    from unboundednerfpytorch_amd import adam_upd_cuda, compat, render_utils_cuda
    if not build_ref.built("nofma"):
        pytest.skip("oracle/_ref not built")
    names = compat.install_as_reference_extensions()
    try:
        for variant in ("nofma", "fma", "nofma", "fma"):
            if build_ref.built(variant):
                mods = build_ref.load(variant)
                assert mods["render_utils_cuda"] is not render_utils_cuda
        assert sys.modules["render_utils_cuda"] is render_utils_cuda
        for mod, fn in ((render_utils_cuda, "raw2alpha"), (render_utils_cuda, "alpha2weight"), (adam_upd_cuda, "masked_adam_upd")):
            f = getattr(mod, fn)
            assert isinstance(f, types.FunctionType) and f.__module__ == mod.__name__, (fn, f)
    finally:
        for n in names:
            sys.modules.pop(n, None)
