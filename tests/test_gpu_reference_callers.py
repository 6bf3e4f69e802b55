"""The reference's OWN Python callers, unchanged, over the HIP drop-in modules on the GPU (VERDICT r1 "What's missing" #3;
north star: "so run_train.py / run_render.py and masked_adam call them unchanged").

`compat.install_as_reference_extensions()` registers this package's modules under the names the reference imports
(render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda); then the reference's model files are imported
from the reference tree -- UNERF_REFERENCE_ROOT / /root/reference when present, else the archive oracle/build_ref.py
staged under oracle/_ref/ (git-ignored, shipped to the GPU box by gpurun) -- and the very generator functions that
produced tests/golden/*.npz on the CPU (tests/golden/gen_golden.py: FourierGrid_model.FourierGridModel.forward in render
and in training mode, FourierGrid_grid.FourierGrid / grid.DenseGrid, dvgo.Raw2Alpha / Alphas2Weights autograd,
masked_adam.MaskedAdam, dvgo.DirectVoxGO.forward, dcvgo.DirectContractedVoxGO.forward, dvgo.get_rays_of_a_view) are
re-run with DEVICE = "cuda".  Their outputs must reproduce the committed goldens: per-ray outputs within 1e-4, survivor
counts within +-2 (threshold flips), Adam bit-exact, gradients to 5e-4 of their scale.
torch_scatter (third-party, absent from the image) is served by the two-line index_add_ stand-in of oracle/install_stubs.
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["ctypes", "pybind"])
def regenerated(golden_dir, request):
    """params: the reference's Python over (a) this package's ctypes mirrors, (b) the NATIVE pybind11 modules of binding/
    (INTEGRATION.md section B built as code) -- the same 18 functions of the same C ABI either way"""
    from oracle import build_ref, install_stubs
    ref_root = build_ref.reference_python_root()
    if ref_root is None:
        pytest.skip("no reference tree and no staged oracle/_ref/reference_py.tar (python oracle/build_ref.py)")
    from unboundednerfpytorch_amd import adam_upd_cuda, compat, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    for m in [k for k in sys.modules if k == "FourierGrid" or k.startswith("FourierGrid.")]:
        del sys.modules[m]
    if request.param == "pybind":
        from binding import build as binding_build
        if not binding_build.available():
            pytest.skip("binding/_build not built (python binding/build.py; __graft_entry__.build() does it)")
        names = compat.install_as_reference_extensions(native=True)
        render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda = [sys.modules[n] for n in (
            "render_utils_cuda", "total_variation_cuda", "ub360_utils_cuda", "adam_upd_cuda")]
        assert render_utils_cuda.__file__.endswith(os.path.join("binding", "_build", "ugrid_render_utils_cuda.so"))
    else:
        names = compat.install_as_reference_extensions()
    assert sys.modules["render_utils_cuda"] is render_utils_cuda and len(names) == 4
    hip = types.SimpleNamespace(render_utils_cuda=render_utils_cuda, total_variation_cuda=total_variation_cuda,
                                ub360_utils_cuda=ub360_utils_cuda, adam_upd_cuda=adam_upd_cuda)
    old = (install_stubs.REFERENCE_ROOT, install_stubs.OPS_BACKEND)
    install_stubs.REFERENCE_ROOT, install_stubs.OPS_BACKEND = ref_root, hip
    import gen_golden
    out = tempfile.mkdtemp(prefix="unerf_regen_")
    gen_golden.DEVICE, gen_golden.OUT = "cuda", out
    # the generators build their inputs with torch.from_numpy and export with .numpy(): route both through the device
    orig_from_numpy, orig_numpy = torch.from_numpy, torch.Tensor.numpy
    torch.from_numpy = lambda a: orig_from_numpy(a).cuda()
    torch.Tensor.numpy = lambda self, *a, **k: orig_numpy(self.detach().cpu(), *a, **k)
    torch.set_default_device("cuda")
    try:
        gen_golden.gen_fouriergrid()
        gen_golden.gen_grid_query()
        gen_golden.gen_autograd_and_adam()
        gen_golden.gen_rays_view()
        gen_golden.gen_dvgo()
        gen_golden.gen_dcvgo()
        gen_golden.gen_train_step()
        # the reference model's forward really went through the HIP library
        import FourierGrid.dvgo as ref_dvgo
        assert ref_dvgo.render_utils_cuda is render_utils_cuda
    finally:
        torch.set_default_device("cpu")
        torch.from_numpy, torch.Tensor.numpy = orig_from_numpy, orig_numpy
        gen_golden.DEVICE, gen_golden.OUT = "cpu", gen_golden.HERE
        install_stubs.REFERENCE_ROOT, install_stubs.OPS_BACKEND = old
        for m in [k for k in sys.modules if k == "FourierGrid" or k.startswith("FourierGrid.")]:
            del sys.modules[m]
        for n in ("render_utils_cuda", "total_variation_cuda", "ub360_utils_cuda", "adam_upd_cuda"):
            sys.modules.pop(n, None)
    yield out, golden_dir
    shutil.rmtree(out, ignore_errors=True)


def load_pair(regenerated, name):
    out, golden_dir = regenerated
    return np.load(os.path.join(out, name + ".npz")), np.load(os.path.join(golden_dir, name + ".npz"))


def check_forward(got, gold, per_ray=("rgb_marched", "depth", "alphainv_last")):
    for k in per_ray:
        if k in gold.files:
            assert got[k].shape == gold[k].shape, k
            assert np.abs(got[k] - gold[k]).max() <= 1e-4, (k, float(np.abs(got[k] - gold[k]).max()))
    assert abs(int(got["weights"].shape[0]) - int(gold["weights"].shape[0])) <= 2            # threshold flips
    if got["weights"].shape == gold["weights"].shape and np.array_equal(got["ray_id"], gold["ray_id"]):
        for k in ("weights", "raw_alpha", "raw_rgb"):
            np.testing.assert_allclose(got[k], gold[k], rtol=0, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("name", ["fg_inf_f3_c12", "fg_l2_f2_c3", "fg_inf_f4_c12_dense", "fg_norgbnet", "fg_inf_f3_c12_medium"])
def test_reference_fouriergrid_model_forward_on_hip_modules(regenerated, name):
    got, gold = load_pair(regenerated, name)
    check_forward(got, gold)
    assert int(got["n_max"]) == int(gold["n_max"])


@pytest.mark.parametrize("name", ["dvgo_fine_direct", "dvgo_fine_residual", "dvgo_coarse", "dcvgo_coarse_l2", "dcvgo_fine_inf"])
def test_reference_dvgo_and_dcvgo_forward_on_hip_modules(regenerated, name):
    got, gold = load_pair(regenerated, name)
    check_forward(got, gold, per_ray=("rgb_marched", "depth", "alphainv_last", "wsum_mid"))


def test_reference_grid_modules_on_hip_modules(regenerated):
    got, gold = load_pair(regenerated, "grid_query")
    for k in gold.files:
        np.testing.assert_allclose(got[k], gold[k], rtol=0, atol=2e-5, err_msg=k)


def test_reference_autograd_functions_and_masked_adam_on_hip_modules(regenerated):
    got, gold = load_pair(regenerated, "autograd_adam")
    np.testing.assert_allclose(got["a2w_alpha"], gold["a2w_alpha"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(got["a2w_w"], gold["a2w_w"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(got["a2w_last"], gold["a2w_last"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(got["a2w_grad_density"], gold["a2w_grad_density"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(got["nonuni_alpha"], gold["nonuni_alpha"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(got["nonuni_grad"], gold["nonuni_grad"], rtol=2e-5, atol=1e-7)
    for k in ("adam_grid", "adam_dense", "adam_grid_m", "adam_grid_v"):       # masked_adam.MaskedAdam: IEEE-only arithmetic
        assert np.array_equal(got[k], gold[k]), k


def test_reference_training_forward_backward_on_hip_modules(regenerated):
    got, gold = load_pair(regenerated, "train_step")
    np.testing.assert_allclose(float(got["loss"]), float(gold["loss"]), rtol=2e-5)
    assert abs(int(got["n_kept"]) - int(gold["n_kept"])) <= 2
    for k in gold.files:
        if k.startswith("grad."):
            scale = float(np.abs(gold[k]).max())
            assert np.abs(got[k] - gold[k]).max() <= 5e-4 * scale + 1e-9, k


def test_reference_get_rays_of_a_view_on_device(regenerated):
    got, gold = load_pair(regenerated, "rays_view")
    for k in gold.files:
        np.testing.assert_allclose(got[k], gold[k], rtol=1e-6, atol=1e-6, err_msg=k)
