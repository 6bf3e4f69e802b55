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


def test_float64_twins_reproduce_reference_kernel_outputs_bit_for_bit(golden_dir):
    """tests/golden/native_ops_f64.npz: outputs of the reference's own kernels called with double tensors on an MI355X
    (tests/golden/gen_native_golden_f64.py).  The fp64 twins must reproduce every one of them exactly, with or without
    oracle/_ref on the box."""
    gold, _ = nc.load_golden(os.path.join(golden_dir, "native_ops_f64.npz"))
    assert gold["raw2alpha"][0].dtype == torch.float64
    got = nc.run_all(hip_modules(), scale=1, device="cuda", chain_from=gold, dtype=torch.float64)
    assert_same(gold, got, "golden float64")


def pybind_modules():
    from binding import build as bb
    if not bb.available():
        pytest.skip("binding/_build not built")
    return bb.load()


@pytest.mark.parametrize("binding", ["ctypes", "pybind"])
@pytest.mark.parametrize("variant", ["nofma", "fma"])
@pytest.mark.parametrize("scale", [1, 8])
def test_float64_twins_vs_live_reference_kernels(variant, scale, binding):
    """The reference's modules dispatch float AND double (AT_DISPATCH_FLOATING_TYPES).  The double instantiations of the drop-in ops
    (include/ugrid_hip_f64.h, libugrid_hip_f64.so) against the reference's own kernels called with double tensors: all 18
    functions, bit for bit -- the reference keeps many intermediates in float whatever the tensor type is (slab distances,
    sample positions, the transmittance, the TV accumulator, the running distance), so "compute in double" would NOT reproduce
    it.  The chained cases (infer_n_samples, raw2alpha_backward, alpha2weight_backward) take the reference's outputs as inputs."""
    from oracle import build_ref
    if not build_ref.built(variant):
        pytest.skip("oracle/_ref/%s not built (python oracle/build_ref.py in the build container)" % variant)
    ref = nc.run_all(build_ref.load(variant), scale=scale, device="cuda", dtype=torch.float64)
    mods = hip_modules() if binding == "ctypes" else pybind_modules()      # both bindings of the C ABI
    got = nc.run_all(mods, scale=scale, device="cuda", chain_from=ref, dtype=torch.float64)
    for name in ref:
        for t in ref[name]:
            assert t.dtype in (torch.float64, torch.int64, torch.bool), (name, t.dtype)
    assert_same(ref, got, "live float64 " + variant)
    assert ref["sample_pts_on_rays"][0].dtype == torch.float64 and ref["sample_pts_on_rays"][0].shape[0] > 1000 * scale


def test_float64_and_mixed_type_errors():
    """one floating type per call; anything but float32 / float64 is refused, as is a double tensor for the ops that have no
    reference counterpart"""
    from unboundednerfpytorch_amd import adam_upd_cuda, render_utils_cuda
    d32 = torch.randn(100, device="cuda")
    with pytest.raises(RuntimeError):
        render_utils_cuda.raw2alpha(d32.half(), -9.0, 0.5)
    with pytest.raises(RuntimeError):
        render_utils_cuda.raw2alpha_nonuni(d32.double(), -9.0, torch.full_like(d32, 0.5))
    with pytest.raises(RuntimeError):
        adam_upd_cuda.masked_adam_upd_rezero(d32.double(), d32.double(), d32.double(), d32.double(), 1, 0.9, 0.99, 0.1, 1e-8)
    e, a = render_utils_cuda.raw2alpha(d32.double(), -9.0, 0.5)
    assert e.dtype == a.dtype == torch.float64


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
