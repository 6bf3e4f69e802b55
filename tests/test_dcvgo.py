"""Contracted-unbounded DVGOv2 (dcvgo.DirectContractedVoxGO, the "dcvgo contracted bg grid" of BASELINE.json
configs[1]): the product's composition DirectContractedVoxGORenderer against golden vectors produced by the
reference's own model class (tests/golden/gen_golden.py::gen_dcvgo).
CPU: the same composition with the oracle's extension modules injected (checks the host logic, all return keys,
bit-exact index outputs).  GPU: the default HIP ops through the C ABI."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import model_oracle, ref_ops


def dcvgo_state(seed, G, Gb, C, norm, dm, ds):
    from unboundednerfpytorch_amd.dcvgo_render import dcvgo_state_from_params
    lo, hi = torch.Tensor([-1, -1, -1]) - 0.2, torch.Tensor([1, 1, 1]) + 0.2
    ws = ((hi - lo) / ((hi - lo).prod() / G ** 3).pow(1 / 3)).long().tolist()
    p = synth.dvgo_params(seed, ws, C, True, dens_mean=dm, dens_std=ds)
    names = ['rgbnet.0', 'rgbnet.2.0', 'rgbnet.3']
    w = [torch.from_numpy(p[n + '.weight']) for n in names] if C > 0 else []
    b = [torch.from_numpy(p[n + '.bias']) for n in names] if C > 0 else []
    st = dcvgo_state_from_params(synth.DCVGO_BOX[0], synth.DCVGO_BOX[1], G ** 3, Gb ** 3, 1e-2,
                                 torch.from_numpy(p['density.grid']), torch.from_numpy(p['k0.grid']), w, b,
                                 torch.from_numpy(p['mask_cache.mask']), 1e-4, contracted_norm=norm)
    return st, ws


def dcvgo_rays(seed, R):
    o, d, v = [torch.from_numpy(a) for a in synth.rays(seed, R, origin_scale=0.5)]
    o = o + torch.tensor(synth.DCVGO_BOX[0]) * 0.5 + torch.tensor(synth.DCVGO_BOX[1]) * 0.5
    return o, d, v


@pytest.mark.parametrize("case", synth.DCVGO_CASES, ids=[c[0] for c in synth.DCVGO_CASES])
def test_dcvgo_composition_matches_reference_golden_cpu(case, golden_dir):
    from unboundednerfpytorch_amd.dcvgo_render import DirectContractedVoxGORenderer
    name, seed, G, Gb, C, norm, R, dm, ds = case
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    torch.set_num_threads(1)
    state, ws = dcvgo_state(seed, G, Gb, C, norm, dm, ds)
    assert ws == gold["world_size"].tolist()
    o, d, v = dcvgo_rays(seed, R)
    rend = DirectContractedVoxGORenderer(state, "cpu", ops=ref_ops, query=model_oracle.fourier_grid_query)
    out = rend(o, d, v, stepsize=0.5, bg=1, render_depth=True)
    assert out["n_max"] == int(gold["n_max"])
    assert np.array_equal(out["ray_id"].numpy(), gold["ray_id"]) and np.array_equal(out["step_id"].numpy(), gold["step_id"])
    for k in ("alphainv_last", "weights", "wsum_mid", "rgb_marched", "raw_density", "raw_alpha", "raw_rgb", "t", "s", "depth"):
        np.testing.assert_allclose(out[k].numpy(), gold[k], rtol=2e-6, atol=2e-7, err_msg=k)
    with pytest.raises(RuntimeError):
        DirectContractedVoxGORenderer(state, "cpu")          # the default ops are the HIP library: no CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("case", synth.DCVGO_CASES, ids=[c[0] for c in synth.DCVGO_CASES])
def test_dcvgo_hip_matches_reference_golden(case, golden_dir):
    from unboundednerfpytorch_amd.dcvgo_render import DirectContractedVoxGORenderer
    name, seed, G, Gb, C, norm, R, dm, ds = case
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    state, _ = dcvgo_state(seed, G, Gb, C, norm, dm, ds)
    o, d, v = [x.cuda() for x in dcvgo_rays(seed, R)]
    out = DirectContractedVoxGORenderer(state, "cuda:0")(o, d, v, stepsize=0.5, bg=1, render_depth=True)
    # cumdist_thres, the mask cache and the scan are bit-exact ops: the kept-sample sets agree unless a 1-ulp alpha
    # difference crosses a threshold
    if out["ray_id"].shape[0] == gold["ray_id"].shape[0]:
        assert np.array_equal(out["ray_id"].cpu().numpy(), gold["ray_id"])
        assert np.array_equal(out["step_id"].cpu().numpy(), gold["step_id"])
    assert abs(out["ray_id"].shape[0] - gold["ray_id"].shape[0]) <= 2
    for k in ("alphainv_last", "rgb_marched", "depth", "wsum_mid"):
        np.testing.assert_allclose(out[k].cpu().numpy(), gold[k], rtol=0, atol=1e-4, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("case", synth.DCVGO_CASES, ids=[c[0] for c in synth.DCVGO_CASES])
def test_dcvgo_fused_render_matches_reference_golden(case, golden_dir):
    """The FUSED DirectContractedVoxGO inference path (ugrid_render_march_dcvgo: sample table with boundary 2, contraction,
    cumdist_thres rule, mask cache, dense-grid lookup, alpha, compositing, wsum_mid in one kernel; then the shade kernel)
    against the golden vectors of the reference's own model class: the four per-ray outputs within 1e-4."""
    from unboundednerfpytorch_amd.dcvgo_render import DirectContractedVoxGORenderer
    name, seed, G, Gb, C, norm, R, dm, ds = case
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    state, _ = dcvgo_state(seed, G, Gb, C, norm, dm, ds)
    o, d, v = [x.cuda() for x in dcvgo_rays(seed, R)]
    rend = DirectContractedVoxGORenderer(state, "cuda:0")
    assert rend.fused_supported()
    out = rend.render_rays(o, d, v, stepsize=0.5, bg=1, render_depth=True)
    assert rend._fused is not None and rend._fused.dc is not None                # it really took the fused kernels
    for k in ("alphainv_last", "rgb_marched", "depth", "wsum_mid"):
        assert out[k].shape == gold[k].shape, k
        np.testing.assert_allclose(out[k].cpu().numpy(), gold[k], rtol=0, atol=1e-4, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("norm,C", [("inf", 12), ("l2", 12), ("inf", 0)])
def test_dcvgo_fused_render_equals_the_composed_forward_at_scale(norm, C):
    """40 000 random rays through a 48^3 model with a non-trivial mask: the fused path against the composition of the
    drop-in kernels (itself pinned on the reference goldens above).  Rays on which a sample sits within rounding of one of
    the hard decisions (cumdist threshold, alpha / weight thresholds, the T < 1e-3 stop) may differ by that sample; all
    others agree to 2e-5, and such flips are rare."""
    from unboundednerfpytorch_amd.dcvgo_render import DirectContractedVoxGORenderer
    state, _ = dcvgo_state(77, 48, 40, C, norm, 2.0, 6.0)
    R = 40_000
    o, d, v = [x.cuda() for x in dcvgo_rays(78, R)]
    rend = DirectContractedVoxGORenderer(state, "cuda:0")
    bg = torch.tensor([0.2, 0.5, 0.9], device="cuda")
    ref = rend(o, d, v, stepsize=0.5, bg=bg, render_depth=True)
    out = rend.render_rays(o, d, v, stepsize=0.5, bg=bg, render_depth=True)
    worst = torch.zeros(R, device="cuda")
    for k in ("alphainv_last", "rgb_marched", "depth", "wsum_mid"):
        e = (out[k] - ref[k]).abs()
        worst = torch.maximum(worst, e.amax(dim=1) if e.dim() == 2 else e)
    frac_bad = float((worst > 2e-5).float().mean())
    print("dcvgo fused vs composed (%s, C=%d): linf %.3e, rays > 2e-5: %.4f %%, terminated %.2f, mean wsum_mid %.3f"
          % (norm, C, float(worst.max()), 100 * frac_bad, float((ref["alphainv_last"] < 1e-3).float().mean()), float(ref["wsum_mid"].mean())))
    assert frac_bad < 2e-3 and float(worst.max()) < 0.3
    assert float(worst.median()) < 1e-6
    # determinism and independence of the ray order
    out2 = rend.render_rays(o, d, v, stepsize=0.5, bg=bg, render_depth=True, ray_order="coherent")
    for k in ("alphainv_last", "rgb_marched", "depth", "wsum_mid"):
        assert torch.equal(out[k], out2[k]), k


@pytest.mark.gpu
def test_dcvgo_render_view_equals_render_rays_on_the_image_rays():
    from unboundednerfpytorch_amd.dcvgo_render import DirectContractedVoxGORenderer
    from unboundednerfpytorch_amd.fourier_render import get_rays_of_a_view
    state, _ = dcvgo_state(81, 40, 40, 12, "inf", 2.0, 6.0)
    rend = DirectContractedVoxGORenderer(state, "cuda:0")
    assert rend.fused_supported()
    H, W = 64, 96
    K = [[90.0, 0, W / 2], [0, 90.0, H / 2], [0, 0, 1]]
    c2w = torch.tensor([[1.0, 0, 0, 0.1], [0, 0.8, 0.6, -0.2], [0, -0.6, 0.8, 0.3]])
    bg = torch.tensor([0.2, 0.5, 0.9], device="cuda")
    img = rend.render_view(H, W, K, c2w, stepsize=0.5, bg=bg, render_depth=True)
    o, d, v = [x.reshape(-1, 3).contiguous().cuda() for x in get_rays_of_a_view(H, W, K, c2w)]
    ref = rend.render_rays(o, d, v, stepsize=0.5, bg=bg, render_depth=True)
    assert set(img) == {"rgb_marched", "depth", "alphainv_last", "wsum_mid"} and img["rgb_marched"].shape == (H, W, 3)
    for k in img:
        assert torch.equal(img[k].reshape(ref[k].shape), ref[k]), k
    # the frame loop over a FRESH renderer (its fused renderer is built inside the loop, before the two view streams start): three views,
    # two in flight, equal bit for bit to the one-stream loop and to render_view
    import numpy as np
    from unboundednerfpytorch_amd.run_render import render_viewpoints
    fresh = DirectContractedVoxGORenderer(state, "cuda:0")
    c2 = c2w.clone()
    c2[:, 3] += torch.tensor([0.05, -0.02, 0.04])
    poses = [c2w.numpy(), c2.numpy(), c2w.numpy()]
    kw = dict(stepsize=0.5, bg=bg, render_depth=True)
    two = render_viewpoints(fresh, poses, [[H, W]] * 3, [K] * 3, kw)
    one = render_viewpoints(fresh, poses, [[H, W]] * 3, [K] * 3, kw, frames_in_flight=1)
    assert all(np.array_equal(a, b) for a, b in zip(two, one))
    assert np.array_equal(two[0][0], img["rgb_marched"].cpu().numpy()) and np.array_equal(two[0][2], two[0][0])
    assert not np.array_equal(two[0][1], two[0][0])
    assert np.array_equal(two[1][0][..., 0], img["depth"].cpu().numpy())


def test_dcvgo_state_from_reference_checkpoint_equals_state_from_params():
    """the checkpoint route (model_kwargs as DirectContractedVoxGO.get_kwargs writes them, numpy bounds included) gives the state
    the constructor-argument route gives, key by key"""
    from unboundednerfpytorch_amd.dcvgo_render import dcvgo_state_from_reference_checkpoint
    G, Gb, C, norm = 20, 18, 12, "l2"
    ref, _ = dcvgo_state(5, G, Gb, C, norm, 2.0, 6.0)
    sd = {"density.grid": ref["density_grid"], "k0.grid": ref["k0_grid"], "mask_cache.mask": ref["mask"],
          "mask_cache.xyz2ijk_scale": ref["xyz2ijk_scale"], "mask_cache.xyz2ijk_shift": ref["xyz2ijk_shift"],
          "rgbnet.0.weight": ref["rgbnet_weights"][0], "rgbnet.0.bias": ref["rgbnet_biases"][0],
          "rgbnet.2.0.weight": ref["rgbnet_weights"][1], "rgbnet.2.0.bias": ref["rgbnet_biases"][1],
          "rgbnet.3.weight": ref["rgbnet_weights"][2], "rgbnet.3.bias": ref["rgbnet_biases"][2]}
    kw = {"xyz_min": np.asarray(synth.DCVGO_BOX[0], dtype=np.float32), "xyz_max": np.asarray(synth.DCVGO_BOX[1], dtype=np.float32),
          "num_voxels": G ** 3, "num_voxels_base": Gb ** 3, "alpha_init": 1e-2, "fast_color_thres": 1e-4, "contracted_norm": norm,
          "density_type": "DenseGrid", "k0_type": "DenseGrid", "rgbnet_dim": C, "viewbase_pe": 4}
    got = dcvgo_state_from_reference_checkpoint({"model_kwargs": kw, "model_state_dict": sd})
    assert set(got) == set(ref)
    for k, v in ref.items():
        if torch.is_tensor(v):
            assert torch.equal(got[k], v), k
        elif isinstance(v, list):
            assert len(got[k]) == len(v) and all(torch.equal(a, b) for a, b in zip(got[k], v)), k
        else:
            assert got[k] == v, k
    with pytest.raises(NotImplementedError):
        dcvgo_state_from_reference_checkpoint({"model_kwargs": dict(kw, density_type="TensoRFGrid"), "model_state_dict": sd})
