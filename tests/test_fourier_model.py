"""unboundednerfpytorch_amd.fourier_model.FourierGridModel (the training-side mirror of the reference's nn.Module,
SURVEY.md section 8 row f2) against golden vectors produced by the reference's own model class:
  * train_step.npz      loss + gradient of every parameter of one training forward/backward,
  * fg_model_utils.npz  state_dict names/shapes, update_occupancy_cache, the TV wrappers, scale_volume_grid, get_kwargs.
The module is device-agnostic Python over the extension-module boundary; here it runs on the CPU with the oracle's
implementation of that boundary injected (`backend=`), which checks all of its host logic; the HIP implementation of
each op behind the boundary is checked op by op in the GPU tests."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import synth
from oracle import model_oracle, ref_ops


def oracle_backend():
    Raw2Alpha, Alphas2Weights = model_oracle.make_autograd_ops(ref_ops)
    return SimpleNamespace(Raw2Alpha=Raw2Alpha, Alphas2Weights=Alphas2Weights, grid_query=model_oracle.fourier_grid_query,
                           total_variation_cuda=ref_ops.total_variation_cuda, render_utils_cuda=ref_ops.render_utils_cuda)


def build(c, G=None):
    from unboundednerfpytorch_amd.fourier_model import FourierGridModel
    G = G or c["G"]
    m = FourierGridModel(xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1], num_voxels_density=G ** 3, num_voxels_base_density=G ** 3,
                         num_voxels_rgb=G ** 3, num_voxels_base_rgb=G ** 3, num_voxels_viewdir=-1, alpha_init=1e-4,
                         fast_color_thres=c["thres"], contracted_norm=c["norm"], fourier_freq_num=c["F"],
                         rgbnet_dim=c["C"], viewbase_pe=c["pe"], backend=oracle_backend())
    params = synth.fouriergrid_params(c["seed"], G, c["F"], c["C"], viewbase_pe=c["pe"], dens_mean=c["dm"], dens_std=c["ds"])
    sd = m.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            assert tuple(sd[k].shape) == tuple(v.shape), k
            sd[k].copy_(torch.from_numpy(v))
    return m


def test_training_forward_backward_matches_the_reference_model(golden_dir):
    c = synth.TRAIN_CASE
    gold = np.load(os.path.join(golden_dir, "train_step.npz"))
    torch.set_num_threads(1)
    m = build(c)
    o, d, v = [torch.from_numpy(a) for a in synth.rays(c["seed"], c["R"])]
    target = torch.from_numpy(synth.uniform(c["seed"] + 5, c["R"] * 3).reshape(c["R"], 3))
    out = m(o, d, v, global_step=1, is_train=True, stepsize=c["stepsize"], render_depth=True)
    assert set(out) == {"alphainv_last", "weights", "rgb_marched", "raw_density", "raw_alpha", "raw_rgb", "ray_id",
                        "step_id", "n_max", "t", "s", "depth"}                     # FourierGrid_model.py:650-672
    loss = torch.nn.functional.mse_loss(out["rgb_marched"], target)
    pout = out["alphainv_last"].clamp(1e-6, 1 - 1e-6)
    loss = loss + 0.01 * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout))).mean()
    loss.backward()
    assert out["weights"].numel() == int(gold["n_kept"])
    np.testing.assert_allclose(float(loss), float(gold["loss"]), rtol=2e-6)
    for name, p in m.named_parameters():
        g = gold["grad." + name]
        scale = np.abs(g).max()
        assert np.abs(p.grad.numpy() - g).max() <= 2e-6 * scale + 1e-12, name
        assert np.array_equal(p.grad.numpy() == 0, g == 0), name


def test_fourier_loss_matches_the_reference_class_and_its_closed_form(golden_dir):
    """weight_freq (bicycle_single.py:57, stump_single.py:55): tests/golden/fourier_loss.npz holds the reference's own FourierMSELoss
    (FourierGrid_model.py:112-129) on seeded colours (a) and one training step of the reference model under bicycle_single's loss
    weights (b).  train_step.fourier_mse_loss is that loss; the closed form the HIP kernels evaluate (csrc/ugrid_train.hip: with
    e = pred - gt, f0 = e0 + e1 + e2, f1 = e0 - (e1 + e2) / 2: mean (f0^2 + 2 f1^2) / 3, gradient 2 / (3 R) [f0 + 2 f1, f0 - f1, f0 - f1])
    reproduces its value and gradient; training_loss with those weights reproduces the step's loss and every parameter gradient."""
    from unboundednerfpytorch_amd import train_step as ts
    gold = np.load(os.path.join(golden_dir, "fourier_loss.npz"))
    R = 257
    pred = torch.from_numpy(synth.uniform(901, R * 3).reshape(R, 3)).requires_grad_(True)
    gt = torch.from_numpy(synth.uniform(902, R * 3).reshape(R, 3))
    val = ts.fourier_mse_loss(pred, gt)
    val.backward()
    np.testing.assert_allclose(float(val), float(gold["a_loss"]), rtol=1e-6)
    np.testing.assert_allclose(pred.grad.numpy(), gold["a_grad"], rtol=1e-5, atol=1e-9)
    e = (pred.detach() - gt).numpy().astype(np.float64)
    f0, f1 = e.sum(1), e[:, 0] - 0.5 * (e[:, 1] + e[:, 2])
    np.testing.assert_allclose((f0 ** 2 + 2 * f1 ** 2).sum() / (3 * R), float(gold["a_loss"]), rtol=1e-6)
    g = 2.0 / (3 * R) * np.stack([f0 + 2 * f1, f0 - f1, f0 - f1], 1)
    np.testing.assert_allclose(g, gold["a_grad"], rtol=1e-5, atol=1e-9)
    # (b) the whole step on the CPU model over the oracle back-end
    c = synth.TRAIN_CASE
    torch.set_num_threads(1)
    m = build(c)
    o, d, v = [torch.from_numpy(a) for a in synth.rays(c["seed"], c["R"])]
    target = torch.from_numpy(synth.uniform(c["seed"] + 5, c["R"] * 3).reshape(c["R"], 3))
    out = m(o, d, v, global_step=1, is_train=True, stepsize=c["stepsize"], render_depth=True)
    cfg = dict(synth.FREQ_WEIGHTS, weight_distortion=0.0, weight_rgbper=0.0)
    loss, mse = ts.training_loss(out, target, cfg, c["R"], near_thres=synth.FREQ_NEAR)
    loss.backward()
    assert out["weights"].numel() == int(gold["b_n_kept"]) and int((out["t"] < synth.FREQ_NEAR).sum()) == int(gold["b_n_near"])
    np.testing.assert_allclose(float(mse), float(gold["b_mse"]), rtol=2e-6)
    np.testing.assert_allclose(float(loss), float(gold["b_loss"]), rtol=2e-6)
    for name, p in m.named_parameters():
        gg = gold["b_grad." + name]
        assert np.abs(p.grad.numpy() - gg).max() <= 2e-6 * np.abs(gg).max() + 1e-12, name


def test_model_utilities_match_the_reference_model(golden_dir):
    c = synth.MODEL_UTILS_CASE
    gold = np.load(os.path.join(golden_dir, "fg_model_utils.npz"))
    torch.set_num_threads(1)
    m = build(c)
    sd = m.state_dict()
    assert sorted(sd.keys()) == gold["sd_keys"].tolist()                             # checkpoints interchange
    assert [str(tuple(sd[k].shape)) for k in sorted(sd.keys())] == gold["sd_shapes"].tolist()
    m.update_occupancy_cache()
    assert np.array_equal(m.mask_cache.mask.numpy(), gold["occ_mask"])
    assert 0.05 < gold["occ_mask"].mean() < 0.95                                     # the case changes the cache
    m.density.grid.grad = torch.from_numpy(synth.normal(700, m.density.grid.numel()).reshape(m.density.grid.shape))
    gk = synth.normal(701, m.k0.grid.numel()).reshape(m.k0.grid.shape)
    gk[np.abs(gk) < 1.0] = 0.0
    m.k0.grid.grad = torch.from_numpy(gk)
    m.density_total_variation_add_grad(1e-3, True)
    m.k0_total_variation_add_grad(2e-3, False)
    assert np.array_equal(m.density.grid.grad.numpy(), gold["tv_density_grad"])
    assert np.array_equal(m.k0.grid.grad.numpy(), gold["tv_k0_grad"])
    m.scale_volume_grid(c["G2"] ** 3, c["G2"] ** 3)
    assert m.world_size_density.tolist() == gold["scaled_world_size"].tolist()
    np.testing.assert_allclose(m.density.grid.detach().numpy(), gold["scaled_density"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(m.k0.grid.detach().numpy(), gold["scaled_k0"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(m.mask_cache.mask.numpy(), gold["scaled_mask"])
    assert abs(float(m.voxel_size_ratio_density) - float(gold["scaled_ratio"])) < 1e-7
    assert sorted(m.get_kwargs().keys()) == gold["kwargs_keys"].tolist()
    # voxel_count_views on the rescaled model (three tiny views inside the unit cube)
    from unboundednerfpytorch_amd.fourier_render import get_rays_of_a_view
    H, W, K, poses = synth.dvgo_views()
    ro, rd = [], []
    for c2w in poses:
        c2w = c2w.copy(); c2w[:, 3] *= 0.25
        o_, d_, _ = get_rays_of_a_view(H, W, K, torch.from_numpy(c2w))
        ro.append(o_); rd.append(d_)
    cnt = m.voxel_count_views(rays_o_tr=torch.stack(ro), rays_d_tr=torch.stack(rd), imsz=1, near=0.05, far=6.0,
                              stepsize=0.5, downrate=1, irregular_shape=False)
    assert cnt.shape == gold["view_count"].shape and float(gold["view_count"].max()) == 3.0
    assert float((cnt.numpy() != gold["view_count"]).mean()) < 2e-3
    # the rescaled model still runs and its new grids receive gradients
    o, d, v = [torch.from_numpy(a) for a in synth.rays(3, 32)]
    out = m(o, d, v, global_step=2, is_train=True, stepsize=0.5)
    out["rgb_marched"].sum().backward()
    assert m.density.grid.grad is not None and tuple(m.density.grid.shape[2:]) == (10, 10, 10)


def test_unsupported_options_fail_loudly():
    from unboundednerfpytorch_amd.fourier_model import FourierGridModel
    kw = dict(xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1], num_voxels_density=512, num_voxels_base_density=512, num_voxels_rgb=512,
              num_voxels_base_rgb=512, alpha_init=1e-4, rgbnet_dim=4, fourier_freq_num=2, backend=oracle_backend())
    with pytest.raises(NotImplementedError):
        FourierGridModel(num_voxels_viewdir=64, **kw)
    with pytest.raises(NotImplementedError):
        FourierGridModel(img_emb_dim=8, sample_num=10, **kw)


@pytest.mark.gpu
def test_training_model_on_hip_matches_the_oracle_backend():
    """The same module with its default back-end (libugrid_hip.so) on the GPU against the CPU run above: loss,
    gradients, occupancy cache and rescaled grids."""
    from unboundednerfpytorch_amd.fourier_model import FourierGridModel
    c = synth.MODEL_UTILS_CASE
    torch.set_num_threads(4)
    ref = build(c)
    dev = FourierGridModel(xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1], num_voxels_density=c["G"] ** 3,
                           num_voxels_base_density=c["G"] ** 3, num_voxels_rgb=c["G"] ** 3, num_voxels_base_rgb=c["G"] ** 3,
                           num_voxels_viewdir=-1, alpha_init=1e-4, fast_color_thres=c["thres"], contracted_norm=c["norm"],
                           fourier_freq_num=c["F"], rgbnet_dim=c["C"], viewbase_pe=c["pe"])
    dev.load_state_dict(ref.state_dict())
    dev = dev.cuda()
    o, d, v = [torch.from_numpy(a) for a in synth.rays(9, 200)]
    target = torch.from_numpy(synth.uniform(10, 600).reshape(200, 3))
    losses = []
    for m, to in ((ref, lambda x: x), (dev, lambda x: x.cuda())):
        out = m(to(o), to(d), to(v), global_step=1, is_train=True, stepsize=0.5, render_depth=True)
        loss = torch.nn.functional.mse_loss(out["rgb_marched"], to(target))
        loss.backward()
        m.density_total_variation_add_grad(1e-3, True)
        losses.append(float(loss.detach()))
    assert abs(losses[0] - losses[1]) <= 1e-5 * max(1.0, abs(losses[0]))
    for (n0, p0), (n1, p1) in zip(ref.named_parameters(), dev.named_parameters()):
        assert n0 == n1
        scale = float(p0.grad.abs().max()) + 1e-12
        assert float((p0.grad - p1.grad.cpu()).abs().max()) <= 5e-4 * scale, n0
    ref.update_occupancy_cache(); dev.update_occupancy_cache()
    assert float((ref.mask_cache.mask != dev.mask_cache.mask.cpu()).float().mean()) < 5e-3
    ref.scale_volume_grid(c["G2"] ** 3, c["G2"] ** 3); dev.scale_volume_grid(c["G2"] ** 3, c["G2"] ** 3)
    np.testing.assert_allclose(dev.k0.grid.detach().cpu().numpy(), ref.k0.grid.detach().numpy(), rtol=1e-5, atol=1e-5)
    assert float((ref.mask_cache.mask != dev.mask_cache.mask.cpu()).float().mean()) < 5e-3
    assert dev.mask_cache.mask.is_cuda and dev.mask_cache.xyz2ijk_scale.is_cuda


TRAIN_CFG = dict(lrate_density=1e-1, lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_viewfreq=0.5, lrate_nosuchfield=1.0,
                 lrate_decay=20, skip_zero_grad_fields=['density', 'k0'])      # == gen_golden.TRAIN_CFG


def test_optimizer_factory_matches_the_reference(golden_dir):
    """create_optimizer_or_freeze_model vs the groups the reference's utils.py:26-56 builds on its own model: same
    order, learning rates (incl. the decay by global_step), skip flags, parameter shapes and frozen parameters."""
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model
    gold = np.load(os.path.join(golden_dir, "train_utils.npz"))
    c = synth.MODEL_UTILS_CASE
    for tag, step, over in (("a", 0, {}), ("b", 5000, {}), ("c", 300, {"lrate_rgbnet": 0.0})):
        m = build(c)
        opt = create_optimizer_or_freeze_model(m, {**TRAIN_CFG, **over}, step, ops=ref_ops)
        assert isinstance(opt, MaskedAdam)
        np.testing.assert_allclose([g["lr"] for g in opt.param_groups], gold[tag + "_lr"], rtol=1e-12)
        assert [bool(g["skip_zero_grad"]) for g in opt.param_groups] == gold[tag + "_skip"].tolist()
        assert [";".join(str(tuple(p.shape)) for p in g["params"]) for g in opt.param_groups] == gold[tag + "_shapes"].tolist()
        frozen = sorted(n for n, p in m.named_parameters() if not p.requires_grad)
        # reference quirk (not mirrored): for an nn.Module field utils.py:53 sets `module.requires_grad = False`, which
        # freezes nothing -- the field is merely left out of the optimizer.  Here its parameters really stop requiring
        # gradients; the optimised parameters and their updates are the same either way.
        assert gold[tag + "_frozen"].tolist() == []
        assert frozen == ([] if tag != "c" else sorted(n for n, _ in m.named_parameters() if n.startswith("rgbnet.")))
    assert int(gold["reverse_checkpoint_ok"]) == 1     # gen_golden: the reference loaded OUR checkpoint and rendered identically


def test_checkpoints_interchange_with_the_reference(golden_dir, tmp_path):
    """forward direction: fg_ckpt_small.tar was written by the reference model class; load_model rebuilds this package's
    model from it and the render matches the reference's (fg_ckpt_small_render.npz).  Then a save / load round trip
    incl. the optimizer state."""
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model, load_checkpoint, load_model, save_checkpoint
    gold = np.load(os.path.join(golden_dir, "fg_ckpt_small_render.npz"))
    torch.set_num_threads(1)
    m, kwargs = load_model(os.path.join(golden_dir, "fg_ckpt_small.tar"), backend=oracle_backend())
    assert kwargs["fourier_freq_num"] == 2 and m.world_len_density == int(gold["world_len"])
    o, d, v = [torch.from_numpy(a) for a in synth.rays(41, 64, origin_scale=0.6)]
    o = o + torch.tensor([0.0, 1.0, -1.0])
    with torch.no_grad():
        out = m(o, d, v, stepsize=0.5, render_depth=True)
    for k in ("rgb_marched", "depth", "alphainv_last"):
        np.testing.assert_allclose(out[k].numpy(), gold[k], rtol=2e-6, atol=2e-7, err_msg=k)
    opt = create_optimizer_or_freeze_model(m, TRAIN_CFG, 0, ops=ref_ops)
    m(o, d, v, global_step=1, is_train=True, stepsize=0.5)["rgb_marched"].sum().backward()
    opt.step()
    path = str(tmp_path / "fine_last.tar")
    save_checkpoint(path, m, opt, 124)
    m2, _ = load_model(path, backend=oracle_backend())
    opt2 = create_optimizer_or_freeze_model(m2, TRAIN_CFG, 0, ops=ref_ops)
    m2, opt2, start = load_checkpoint(m2, opt2, path, no_reload_optimizer=False)
    assert start == 124
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[k]["exp_avg_sq"], s2[k]["exp_avg_sq"]) for k in s1)


def test_train_iteration_schedule_and_losses():
    """train_step.py on the CPU (oracle back-end): the loss equals a hand composition of its terms, total variation only
    acts inside its step window, progressive scaling rebuilds grids + optimizer and lowers act_shift, and a short run
    reduces the photometric error."""
    from unboundednerfpytorch_amd import train_step as ts
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model
    torch.set_num_threads(2)
    c = dict(synth.MODEL_UTILS_CASE, dm=0.5, ds=2.0)
    m = build(c)
    cfg_model = dict(num_voxels_density=12 ** 3, num_voxels_rgb=12 ** 3)
    cfg = dict(TRAIN_CFG, weight_main=1.0, weight_freq=0.1, weight_entropy_last=0.001, weight_distortion=0.001, weight_rgbper=0.05,
               weight_tv_density=1e-4, weight_tv_k0=1e-5, tv_after=0, tv_before=4, tv_every=1, tv_dense_before=3,
               pg_scale=[3, 5], decay_after_scale=0.25)
    dist_fn = model_oracle_distortion()
    o, d, v = [torch.from_numpy(a) for a in synth.rays(77, 128)]
    target = torch.sigmoid(torch.from_numpy(synth.normal(78, 128 * 3).reshape(128, 3)))
    rk = dict(stepsize=0.5, render_depth=False)
    opt = create_optimizer_or_freeze_model(m, cfg, 0, ops=ref_ops)
    # loss = hand composition
    out = m(o, d, v, global_step=1, is_train=True, **rk)
    loss, mse = ts.training_loss(out, target, cfg, 128, distortion_fn=dist_fn)
    p = out['alphainv_last'].clamp(1e-6, 1 - 1e-6)
    per = (out['raw_rgb'] - target[out['ray_id']]).pow(2).sum(-1)
    want = (mse + 0.1 * ts.fourier_mse_loss(out['rgb_marched'], target)
            + 0.001 * (-(p * torch.log(p) + (1 - p) * torch.log(1 - p))).mean()
            + 0.001 * dist_fn(out['weights'], out['s'], 1 / out['n_max'], out['ray_id'])
            + 0.05 * (per * out['weights'].detach()).sum() / 128)
    assert abs(float(loss) - float(want)) <= 1e-6 * max(1.0, abs(float(want)))
    # schedule
    calls, cur = [], {"step": 0}

    class _RecordingTV:      # the TV term reaches the density grid through its tv_module (train_step's grad_hook)
        @staticmethod
        def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
            calls.append((cur["step"], dense_mode))
            return ref_ops.total_variation_cuda.total_variation_add_grad(param, grad, wx, wy, wz, dense_mode)
    m.density.tv_module = _RecordingTV
    shifts, sizes, psnrs = [], [], []
    for step in range(1, 9):
        cur["step"] = step
        opt = ts.maybe_scale_grids(m, opt, cfg, cfg_model, step, ops=ref_ops)
        lr_before = opt.param_groups[0]['lr']
        loss, psnr = ts.train_iteration(m, opt, o, d, v, target, cfg, step, rk, distortion_fn=dist_fn)
        assert abs(opt.param_groups[0]['lr'] / lr_before - 0.1 ** (1 / 20000)) < 1e-12
        shifts.append(float(m.act_shift)); sizes.append(tuple(m.density.grid.shape[2:])); psnrs.append(psnr)
    assert [s for s, _ in calls] == [1, 2, 3] and [dm for _, dm in calls] == [True, True, False]      # 0 < step < 4, dense < 3
    assert sizes[1] == (8, 8, 8) and sizes[2] == (9, 9, 9) and sizes[4] == (12, 12, 12)               # 12^3/2 -> 9^3, then 12^3
    assert abs(shifts[1] - shifts[2] - 0.25) < 1e-6 and abs(shifts[3] - shifts[4] - 0.25) < 1e-6
    # after the schedule the optimisation makes steady progress on the (random) target
    for step in range(9, 40):
        cur["step"] = step
        psnrs.append(ts.train_iteration(m, opt, o, d, v, target, cfg, step, rk, distortion_fn=dist_fn)[1])
    assert psnrs[-1] > psnrs[7] + 0.5, (psnrs[7], psnrs[-1])


def model_oracle_distortion():
    """flatten_eff_distloss(w, m, interval, ray_id) (the library call of run_train.py:274) over the oracle's
    segment_cumsum: the product class with its test hook pointing at the oracle op."""
    from unboundednerfpytorch_amd import ops

    def fn(w, s, interval, ray_id):
        ops.DistortionLoss.segment_cumsum = staticmethod(
            lambda w_, s_, r_, n_: ref_ops.segment_cumsum(w_.detach().contiguous(), s_.contiguous(), r_.contiguous(), n_))
        try:
            return ops.flatten_eff_distloss(w, s, interval, ray_id)
        finally:
            ops.DistortionLoss.segment_cumsum = None
    return fn


def test_flatten_eff_distloss_backward_is_the_derivative_of_its_forward():
    """the training loop's distortion term is torch_efficient_distloss.flatten_eff_distloss
    (run_train.py:274), whose backward divides by n_rays; the reference's in-repo DistortionLoss (dead code) does
    not.  ops.flatten_eff_distloss must be the derivative of its own value: checked against torch autograd of the
    definition in fp64, and against the quirky class (ratio exactly n_rays)."""
    from unboundednerfpytorch_amd import ops
    w, s, ray_id, n_max = synth.distortion_inputs()
    R = int(ray_id.max()) + 1
    cum = staticmethod(lambda w_, s_, r_, n_: ref_ops.segment_cumsum(w_.detach().contiguous(), s_.contiguous(), r_.contiguous(), n_))
    ops.DistortionLoss.segment_cumsum = cum
    try:
        wt = torch.from_numpy(w).requires_grad_(True)
        loss = ops.flatten_eff_distloss(wt, torch.from_numpy(s), 1 / n_max, torch.from_numpy(ray_id))
        loss.backward()
        g_new = wt.grad.clone()
        wq = torch.from_numpy(w).requires_grad_(True)
        loss_q = ops.distortion_loss(wq, torch.from_numpy(s), n_max, torch.from_numpy(ray_id))
        loss_q.backward()
    finally:
        ops.DistortionLoss.segment_cumsum = None
    assert float(loss) == float(loss_q)
    np.testing.assert_allclose(g_new.numpy() * R, wq.grad.numpy(), rtol=1e-6, atol=1e-9)
    # definition, fp64 autograd
    w64 = torch.from_numpy(w.astype(np.float64)).requires_grad_(True)
    s64 = torch.from_numpy(s.astype(np.float64))
    total = 0.0
    for r in range(R):
        m = torch.from_numpy(ray_id == r)
        if int(m.sum()) == 0:
            continue
        wr, sr = w64[m], s64[m]
        total = total + (wr[:, None] * wr[None, :] * (sr[:, None] - sr[None, :]).abs()).sum() + (wr ** 2).sum() / (3 * n_max)
    (total / R).backward()
    np.testing.assert_allclose(float(loss), float(total / R), rtol=2e-5)
    np.testing.assert_allclose(g_new.numpy(), w64.grad.numpy(), rtol=2e-4, atol=2e-7)
