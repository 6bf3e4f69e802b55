"""Generate tests/golden/*.npz by running the REFERENCE's own Python model code.

Runs only in the build container (needs /root/reference).  The reference's four CUDA extension
modules are replaced by the C oracle through oracle/install_stubs.py (the reference has no CPU
path of its own); everything else executed is the reference's code:
FourierGrid_model.FourierGridModel / FourierGrid_grid.FourierGrid / grid.DenseGrid /
dvgo.Raw2Alpha, Alphas2Weights / masked_adam.MaskedAdam.

Inputs are regenerated from seeds (tests/synth.py); the fixtures hold reference OUTPUTS only.

    python tests/golden/gen_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# tests/test_gpu_reference_callers.py re-runs these generators with the HIP modules installed as the reference's
# extension modules, DEVICE = "cuda" and OUT = a scratch directory, and compares the files with the committed ones
DEVICE = "cpu"
OUT = HERE
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth  # noqa: E402
from oracle import install_stubs  # noqa: E402

# (name, seed, G, F, C, viewbase_pe, contracted_norm, stepsize, R, thres, dens_mean, dens_std)
FG_CASES = [
    ("fg_inf_f3_c12", 11, 16, 3, 12, 4, "inf", 0.5, 96, 1e-4, -3.0, 8.0),
    ("fg_l2_f2_c3", 12, 12, 2, 3, 2, "l2", 0.7, 64, 1e-4, -2.0, 6.0),
    ("fg_inf_f4_c12_dense", 13, 10, 4, 12, 4, "inf", 0.5, 48, 1e-4, 8.0, 12.0),
    ("fg_norgbnet", 14, 12, 3, 0, 4, "inf", 0.5, 64, 1e-4, 5.0, 12.0),
    ("fg_inf_f3_c12_medium", 15, 14, 3, 12, 4, "inf", 1.31, 80, 1e-4, 4.0, 12.0),
]
# NB: fast_color_thres == 0 is not a usable mode of the reference model: forward() then hands the
# 2-D [R,S] alpha to alpha2weight and crashes on `weights * s` (FourierGrid_model.py:600-614,667).


def build_reference_model(mod, G, F, C, viewbase_pe, norm, thres, params):
    model = mod.FourierGridModel(
        xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1],
        num_voxels_density=G ** 3, num_voxels_base_density=G ** 3,
        num_voxels_rgb=G ** 3, num_voxels_base_rgb=G ** 3, num_voxels_viewdir=-1,
        alpha_init=1e-4, fast_color_thres=thres, contracted_norm=norm,
        fourier_freq_num=F, rgbnet_dim=C, viewbase_pe=viewbase_pe)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(v))
    assert int(model.world_len_density) == G
    return model.to(DEVICE)


def gen_fouriergrid():
    mod = install_stubs.import_reference("FourierGrid_model")
    for name, seed, G, F, C, pe, norm, stepsize, R, thres, dm, ds in FG_CASES:
        params = synth.fouriergrid_params(seed, G, F, C, viewbase_pe=pe, dens_mean=dm, dens_std=ds)
        model = build_reference_model(mod, G, F, C, pe, norm, thres, params)
        o, d, v = [torch.from_numpy(a) for a in synth.rays(seed, R)]
        with torch.no_grad():
            out = model(o, d, v, stepsize=stepsize, render_depth=True)
        keep = {k: out[k].numpy() for k in ("alphainv_last", "weights", "rgb_marched", "raw_density", "raw_alpha",
                                            "raw_rgb", "ray_id", "step_id", "t", "s", "depth")}
        keep["n_max"] = np.int64(out["n_max"])
        keep["interval"] = np.float32(float(stepsize * model.voxel_size_ratio_density))
        keep["act_shift"] = model.act_shift.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **keep)
        print(name, "M=%d" % out["weights"].numel(), "rgb max %.4f" % float(out["rgb_marched"].max()),
              "terminated %d/%d" % (int((out["alphainv_last"] < 1e-3).sum()), R))


def gen_grid_query():
    """FourierGrid_grid.FourierGrid.forward and grid.DenseGrid.forward on points in and out of bounds."""
    fg = install_stubs.import_reference("FourierGrid_grid")
    dg = install_stubs.import_reference("grid")
    res = {}
    n = 257
    pts = torch.from_numpy(synth.uniform(31, n * 3, -1.5, 1.5).reshape(n, 3))
    pts[:8] = torch.tensor([[-1.2, -1.2, -1.2], [1.2, 1.2, 1.2], [0, 0, 0], [1.2, -1.2, 0.3],
                            [1.3, 0, 0], [0, -1.25, 0], [0.1, 0.2, 1.2000001], [-1.2, 1.2, -1.2]])
    for C, F in ((1, 3), (12, 3), (3, 2)):
        G = (9, 7, 5)
        m = fg.FourierGrid(channels=C, world_size=torch.tensor(G), xyz_min=[-1.2] * 3, xyz_max=[1.2] * 3,
                           use_nerf_pos=True, fourier_freq_num=F, config={}).to(DEVICE)
        g = synth.normal(40 + C, (1 + 2 * F) * C * G[0] * G[1] * G[2]).reshape(1 + 2 * F, C, *G)
        with torch.no_grad():
            m.grid.copy_(torch.from_numpy(g))
            res["fourier_c%d_f%d" % (C, F)] = m(pts).numpy()
    for C in (1, 4):
        G = (6, 8, 11)
        m = dg.DenseGrid(channels=C, world_size=torch.tensor(G), xyz_min=[-1.0, -0.5, -2.0], xyz_max=[1.0, 1.5, 1.0]).to(DEVICE)
        g = synth.normal(50 + C, C * G[0] * G[1] * G[2]).reshape(1, C, *G)
        with torch.no_grad():
            m.grid.copy_(torch.from_numpy(g))
            res["dense_c%d" % C] = m(pts).numpy()
    np.savez_compressed(os.path.join(OUT, "grid_query.npz"), **res)
    print("grid_query", {k: v.shape for k, v in res.items()})


def gen_autograd_and_adam():
    """dvgo.Raw2Alpha / Raw2Alpha_nonuni / Alphas2Weights forward+backward and masked_adam.MaskedAdam
    (host-side dispatch logic of the reference, kernels = C oracle)."""
    dvgo = install_stubs.import_reference("dvgo")
    madam = install_stubs.import_reference("masked_adam")
    res = {}
    n, R = 300, 17
    dens = torch.from_numpy(synth.normal(61, n, 5.0, 6.0)).requires_grad_(True)
    shift = torch.tensor([-9.21024])
    alpha = dvgo.Raw2Alpha.apply(dens, shift, 0.5)
    ray_id = torch.from_numpy(np.sort((synth.uniform(62, n) * R).astype(np.int64)))
    w, last = dvgo.Alphas2Weights.apply(alpha, ray_id, R)
    gw = torch.from_numpy(synth.normal(63, n))
    gl = torch.from_numpy(synth.normal(64, R))
    (w * gw).sum().add((last * gl).sum()).backward()
    res.update(a2w_alpha=alpha.detach().numpy(), a2w_w=w.detach().numpy(), a2w_last=last.detach().numpy(),
               a2w_grad_density=dens.grad.numpy(), a2w_ray_id=ray_id.numpy())
    dens2 = torch.from_numpy(synth.normal(65, n, 5.0, 6.0)).requires_grad_(True)
    itv = torch.from_numpy(synth.uniform(66, n, 0.1, 1.0))
    a2 = dvgo.Raw2Alpha_nonuni.apply(dens2, shift, itv)
    (a2 * gw).sum().backward()
    res.update(nonuni_alpha=a2.detach().numpy(), nonuni_grad=dens2.grad.numpy())

    # MaskedAdam: 3 steps over a masked grid param, a dense param and a per-voxel-lr param
    shape = (1, 2, 4, 5, 6)
    p_grid = torch.nn.Parameter(torch.from_numpy(synth.normal(70, 240).reshape(shape)))
    p_dense = torch.nn.Parameter(torch.from_numpy(synth.normal(71, 33)))
    opt = madam.MaskedAdam([
        {'params': [p_grid], 'lr': 0.1, 'skip_zero_grad': True},
        {'params': [p_dense], 'lr': 1e-3, 'skip_zero_grad': False}])
    for step in range(3):
        g = synth.normal(80 + step, 240).reshape(shape)
        g[np.abs(g) < 0.8] = 0.0
        p_grid.grad = torch.from_numpy(g)
        p_dense.grad = torch.from_numpy(synth.normal(90 + step, 33))
        if step == 2:
            opt.set_pervoxel_lr(torch.from_numpy(synth.uniform(95, 240, 0.0, 9.0).reshape(shape)).floor())
        opt.step()
    res.update(adam_grid=p_grid.detach().numpy(), adam_dense=p_dense.detach().numpy(),
               adam_grid_m=opt.state[p_grid]['exp_avg'].numpy(), adam_grid_v=opt.state[p_grid]['exp_avg_sq'].numpy())
    np.savez_compressed(os.path.join(OUT, "autograd_adam.npz"), **res)
    print("autograd_adam", {k: v.shape for k, v in res.items()})


def gen_distortion():
    """The reference's own DistortionLoss class (FourierGrid_model.py:684-708) on CPU; its segment_cumsum call is
    served by the oracle op (the reference extension never exported one)."""
    fgm = install_stubs.import_reference("FourierGrid_model")
    w, s, ray_id, n_max = synth.distortion_inputs()
    wt = torch.from_numpy(w).requires_grad_(True)
    loss = fgm.distortion_loss(wt, torch.from_numpy(s), n_max, torch.from_numpy(ray_id))
    loss.backward()
    from oracle import ref_ops
    pre = ref_ops.segment_cumsum(torch.from_numpy(w), torch.from_numpy(s), torch.from_numpy(ray_id))
    res = dict(loss=loss.detach().numpy(), grad=wt.grad.numpy(), w_prefix=pre[0].numpy(), w_total=pre[1].numpy(),
               ws_prefix=pre[2].numpy(), ws_total=pre[3].numpy())
    np.savez_compressed(os.path.join(OUT, "distortion.npz"), **res)
    print("distortion", float(loss), wt.grad.shape)


def gen_train_step():
    """One training forward + backward of the reference FourierGridModel itself (forward(..., global_step=1,
    is_train=True), loss = mse + 0.01 * entropy_last as run_train.py:254-261 forms it): the gradients of every
    parameter, to pin oracle/model_oracle.fouriergrid_train_forward (the CPU side of tests/test_gpu_train_step.py)."""
    mod = install_stubs.import_reference("FourierGrid_model")
    c = synth.TRAIN_CASE
    params = synth.fouriergrid_params(c["seed"], c["G"], c["F"], c["C"], viewbase_pe=c["pe"], dens_mean=c["dm"], dens_std=c["ds"])
    model = build_reference_model(mod, c["G"], c["F"], c["C"], c["pe"], c["norm"], c["thres"], params)
    o, d, v = [torch.from_numpy(a) for a in synth.rays(c["seed"], c["R"])]
    target = torch.from_numpy(synth.uniform(c["seed"] + 5, c["R"] * 3).reshape(c["R"], 3))
    out = model(o, d, v, global_step=1, is_train=True, stepsize=c["stepsize"], render_depth=True)
    loss = torch.nn.functional.mse_loss(out["rgb_marched"], target)
    pout = out["alphainv_last"].clamp(1e-6, 1 - 1e-6)
    loss = loss + 0.01 * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout))).mean()
    loss.backward()
    res = {"loss": loss.detach().numpy(), "n_kept": np.int64(out["weights"].numel()),
           "rgb_marched": out["rgb_marched"].detach().numpy(), "alphainv_last": out["alphainv_last"].detach().numpy()}
    for k, p in model.named_parameters():
        if p.grad is not None:
            res["grad." + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **res)
    print("train_step loss %.6f kept %d grads %s" % (float(loss), int(res["n_kept"]), sorted(k for k in res if k.startswith("grad."))))


def gen_rays_view():
    """dvgo.get_rays_of_a_view (dvgo.py:493-521,554-559) for a small view and three flag combinations."""
    dvgo = install_stubs.import_reference("dvgo")
    K = np.array([[9.0, 0, 3.5], [0, 8.0, 2.5], [0, 0, 1]], dtype=np.float32)
    ang = 0.7
    c2w = np.array([[np.cos(ang), 0, np.sin(ang), 0.3], [0.1, 1.0, 0, -0.2], [-np.sin(ang), 0, np.cos(ang), 0.4]],
                   dtype=np.float32)
    res = {"K": K, "c2w": c2w}
    for tag, kw in (("a", dict(inverse_y=False, flip_x=False, flip_y=False)),
                    ("b", dict(inverse_y=True, flip_x=True, flip_y=False)),
                    ("c", dict(inverse_y=False, flip_x=False, flip_y=True))):
        o, d, v = dvgo.get_rays_of_a_view(H=5, W=7, K=K, c2w=torch.from_numpy(c2w), ndc=False, mode='center', **kw)
        res[tag + "_o"], res[tag + "_d"], res[tag + "_v"] = o.numpy(), d.numpy(), v.numpy()
    np.savez_compressed(os.path.join(OUT, "rays_view.npz"), **res)
    print("rays_view", o.shape)


DVGO_CASES = [
    # name, seed, G, C (0 = coarse stage), rgbnet_direct, R, dens_mean, dens_std
    ("dvgo_fine_direct", 31, 22, 12, True, 150, 2.0, 4.0),
    ("dvgo_fine_residual", 32, 18, 9, False, 120, 3.0, 5.0),
    ("dvgo_coarse", 33, 20, 0, False, 120, 1.0, 4.0),
]


def dvgo_inputs(seed, G, C):
    """Synthetic DirectVoxGO parameters + a random occupancy mask + rays around a [-1,1]x[-0.8,0.9]x[-1.1,1] box."""
    xyz_min, xyz_max = [-1.0, -0.8, -1.1], [1.0, 0.9, 1.0]
    nvox = G ** 3
    return xyz_min, xyz_max, nvox


def gen_dvgo():
    """dvgo.DirectVoxGO.forward (bounded scenes, config 1): variable-length sampling, mask cache, dense grids."""
    dvgo = install_stubs.import_reference("dvgo")
    for name, seed, G, C, direct, R, dm, ds in DVGO_CASES:
        xyz_min, xyz_max, nvox = dvgo_inputs(seed, G, C)
        model = dvgo.DirectVoxGO(xyz_min=xyz_min, xyz_max=xyz_max, num_voxels=nvox, num_voxels_base=nvox,
                                 alpha_init=1e-2, fast_color_thres=1e-4, rgbnet_dim=C, rgbnet_direct=direct,
                                 mask_cache_world_size=None)
        ws = [int(x) for x in model.world_size]
        sd = model.state_dict()
        params = synth.dvgo_params(seed, ws, C, direct, dens_mean=dm, dens_std=ds)
        with torch.no_grad():
            for k, v in params.items():
                assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
                sd[k].copy_(torch.from_numpy(v))
        model = model.to(DEVICE)
        o, d, v = [torch.from_numpy(a) for a in synth.rays(seed, R, origin_scale=0.4)]
        with torch.no_grad():
            out = model(o, d, v, near=0.2, far=6.0, stepsize=0.5, bg=1, render_depth=True)
        keep = {k: out[k].numpy() for k in ("alphainv_last", "weights", "rgb_marched", "raw_alpha", "raw_rgb", "ray_id", "depth")}
        keep["world_size"] = np.array(ws)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **keep)
        print(name, "world", ws, "M=%d" % out["weights"].numel(), "rgb mean %.3f" % float(out["rgb_marched"].mean()))


def gen_dvgo_utils():
    """Training-ray preparation of the bounded model, from the reference's own methods: DirectVoxGO.hit_coarse_geo
    (dvgo.py:292-304), DirectVoxGO.voxel_count_views (:247-277) and get_training_rays_in_maskcache_sampling (:619-657)
    on three tiny views of the dvgo_fine_direct model."""
    dvgo = install_stubs.import_reference("dvgo")
    name, seed, G, C, direct, R, dm, ds = DVGO_CASES[0]
    xyz_min, xyz_max, nvox = dvgo_inputs(seed, G, C)
    model = dvgo.DirectVoxGO(xyz_min=xyz_min, xyz_max=xyz_max, num_voxels=nvox, num_voxels_base=nvox, alpha_init=1e-2,
                             fast_color_thres=1e-4, rgbnet_dim=C, rgbnet_direct=direct, mask_cache_world_size=None)
    ws = [int(x) for x in model.world_size]
    sd = model.state_dict()
    params = synth.dvgo_params(seed, ws, C, direct, dens_mean=dm, dens_std=ds)
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(torch.from_numpy(v))
    H, W, K, poses = synth.dvgo_views()
    imgs = [torch.from_numpy(synth.uniform(900 + i, H * W * 3).reshape(H, W, 3)) for i in range(len(poses))]
    rk = dict(near=0.2, far=6.0, stepsize=0.5)
    ro, rd = [], []
    hits = []
    for c2w in poses:
        o, d, _ = dvgo.get_rays_of_a_view(H=H, W=W, K=K, c2w=torch.from_numpy(c2w), ndc=False, inverse_y=False,
                                          flip_x=False, flip_y=False)
        ro.append(o); rd.append(d)
        hits.append(model.hit_coarse_geo(rays_o=o, rays_d=d, **rk).numpy())
    count = model.voxel_count_views(rays_o_tr=torch.stack(ro), rays_d_tr=torch.stack(rd), imsz=1, near=0.2, far=6.0,
                                    stepsize=0.5, downrate=1, irregular_shape=False)
    rgb_tr, o_tr, d_tr, v_tr, imsz = dvgo.get_training_rays_in_maskcache_sampling(
        rgb_tr_ori=imgs, train_poses=[torch.from_numpy(p) for p in poses], HW=[(H, W)] * len(poses), Ks=[K] * len(poses),
        ndc=False, inverse_y=False, flip_x=False, flip_y=False, model=model, render_kwargs=rk)
    res = dict(hit=np.stack(hits), count=count.detach().numpy(), rgb_tr=rgb_tr.numpy(), rays_o_tr=o_tr.numpy(),
               rays_d_tr=d_tr.numpy(), viewdirs_tr=v_tr.numpy(), imsz=np.array([int(x) for x in imsz]))
    np.savez_compressed(os.path.join(OUT, "dvgo_utils.npz"), **res)
    print("dvgo_utils hit frac %.2f, seen voxels %d of %d, kept rays %s" % (res["hit"].mean(), int((res["count"] > 0).sum()),
                                                                         res["count"].size, res["imsz"].tolist()))


def gen_model_utils():
    """Model-level training utilities of the reference FourierGridModel on the MODEL_UTILS_CASE model: parameter / buffer
    names and shapes, update_occupancy_cache, the TV wrappers, scale_volume_grid (coarse-to-fine resampling + mask
    cache rebuild) and get_kwargs -- to pin unboundednerfpytorch_amd.fourier_model.FourierGridModel."""
    mod = install_stubs.import_reference("FourierGrid_model")
    c = synth.MODEL_UTILS_CASE
    params = synth.fouriergrid_params(c["seed"], c["G"], c["F"], c["C"], viewbase_pe=c["pe"], dens_mean=c["dm"], dens_std=c["ds"])
    model = build_reference_model(mod, c["G"], c["F"], c["C"], c["pe"], c["norm"], c["thres"], params)
    res = {}
    sd = model.state_dict()
    res["sd_keys"] = np.array(sorted(sd.keys()))
    res["sd_shapes"] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])
    model.update_occupancy_cache()
    res["occ_mask"] = model.mask_cache.mask.numpy().copy()
    model.density.grid.grad = torch.from_numpy(synth.normal(700, model.density.grid.numel()).reshape(model.density.grid.shape))
    gk = synth.normal(701, model.k0.grid.numel()).reshape(model.k0.grid.shape)
    gk[np.abs(gk) < 1.0] = 0.0
    model.k0.grid.grad = torch.from_numpy(gk)
    model.density_total_variation_add_grad(1e-3, True)
    model.k0_total_variation_add_grad(2e-3, False)
    res["tv_density_grad"] = model.density.grid.grad.numpy().copy()
    res["tv_k0_grad"] = model.k0.grid.grad.numpy().copy()
    model.scale_volume_grid(c["G2"] ** 3, c["G2"] ** 3)
    res["scaled_density"] = model.density.grid.detach().numpy().copy()
    res["scaled_k0"] = model.k0.grid.detach().numpy().copy()
    res["scaled_mask"] = model.mask_cache.mask.numpy().copy()
    res["scaled_world_size"] = model.world_size_density.numpy().copy()
    res["scaled_ratio"] = np.float32(float(model.voxel_size_ratio_density))
    kw = model.get_kwargs()
    res["kwargs_keys"] = np.array(sorted(kw.keys()))
    # voxel_count_views on the rescaled model: three tiny views placed inside the unit cube
    dvgo = install_stubs.import_reference("dvgo")
    H, W, K, poses = synth.dvgo_views()
    ro, rd = [], []
    for c2w in poses:
        c2w = c2w.copy(); c2w[:, 3] *= 0.25
        o, d, _ = dvgo.get_rays_of_a_view(H=H, W=W, K=K, c2w=torch.from_numpy(c2w), ndc=False, inverse_y=False,
                                          flip_x=False, flip_y=False)
        ro.append(o); rd.append(d)
    cnt = model.voxel_count_views(rays_o_tr=torch.stack(ro), rays_d_tr=torch.stack(rd), imsz=1, near=0.05, far=6.0,
                                  stepsize=0.5, downrate=1, irregular_shape=False)
    res["view_count"] = cnt.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "fg_model_utils.npz"), **res)
    print("model_utils: occupancy %.3f -> scaled world %s mask %.3f ratio %.4f" % (
        res["occ_mask"].mean(), res["scaled_world_size"].tolist(), res["scaled_mask"].mean(), float(res["scaled_ratio"])))


class _Cfg(dict):
    """dict with attribute access, like the mmengine Config the reference passes around"""
    __getattr__ = dict.__getitem__


TRAIN_CFG = dict(lrate_density=1e-1, lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_viewfreq=0.5, lrate_nosuchfield=1.0,
                 lrate_decay=20, skip_zero_grad_fields=['density', 'k0'])


def gen_train_utils():
    """utils.create_optimizer_or_freeze_model (utils.py:26-56) on the reference model: the param groups it builds (lr,
    skip flag, parameter shapes) at two global steps and with a frozen field.  Also the reverse direction of the
    checkpoint interchange: a checkpoint written by THIS package's save_checkpoint from its own FourierGridModel is
    loaded by the reference's load_model and must render identically (checked here, recorded as a flag)."""
    mod = install_stubs.import_reference("FourierGrid_model")
    utils = install_stubs.import_reference("utils")
    c = synth.MODEL_UTILS_CASE
    params = synth.fouriergrid_params(c["seed"], c["G"], c["F"], c["C"], viewbase_pe=c["pe"], dens_mean=c["dm"], dens_std=c["ds"])
    res = {}
    for tag, step, over in (("a", 0, {}), ("b", 5000, {}), ("c", 300, {"lrate_rgbnet": 0.0})):
        model = build_reference_model(mod, c["G"], c["F"], c["C"], c["pe"], c["norm"], c["thres"], params)
        opt = utils.create_optimizer_or_freeze_model(model, _Cfg({**TRAIN_CFG, **over}), global_step=step)
        res[tag + "_lr"] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
        res[tag + "_skip"] = np.array([bool(g["skip_zero_grad"]) for g in opt.param_groups])
        res[tag + "_shapes"] = np.array([";".join(str(tuple(p.shape)) for p in g["params"]) for g in opt.param_groups])
        res[tag + "_frozen"] = np.array(sorted(n for n, p in model.named_parameters() if not p.requires_grad))
    # reverse checkpoint direction
    import tempfile
    from types import SimpleNamespace
    from oracle import model_oracle, ref_ops
    from unboundednerfpytorch_amd.fourier_model import FourierGridModel as Mine
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model, save_checkpoint
    R2A, A2W = model_oracle.make_autograd_ops(ref_ops)
    be = SimpleNamespace(Raw2Alpha=R2A, Alphas2Weights=A2W, grid_query=model_oracle.fourier_grid_query,
                         total_variation_cuda=ref_ops.total_variation_cuda, render_utils_cuda=ref_ops.render_utils_cuda)
    mine = Mine(xyz_min=[-2.0, -1.0, -3.0], xyz_max=[2.0, 3.0, 1.0], num_voxels_density=9 ** 3, num_voxels_base_density=12 ** 3,
                num_voxels_rgb=9 ** 3, num_voxels_base_rgb=12 ** 3, num_voxels_viewdir=-1, alpha_init=1e-3,
                fast_color_thres=1e-4, fourier_freq_num=2, rgbnet_dim=12, backend=be)
    p2 = synth.fouriergrid_params(41, int(mine.world_len_density), 2, 12, dens_mean=4.0, dens_std=10.0)
    sd = mine.state_dict()
    with torch.no_grad():
        for k, v in p2.items():
            sd[k].copy_(torch.from_numpy(v))
    opt = create_optimizer_or_freeze_model(mine, TRAIN_CFG, 0, ops=ref_ops)
    path = os.path.join(tempfile.mkdtemp(), "mine_last.tar")
    save_checkpoint(path, mine, opt, 77)
    import functools
    orig_load = torch.load        # the reference calls torch.load(path) (torch 1.13 semantics: full unpickling)
    torch.load = functools.partial(orig_load, weights_only=False)
    try:
        theirs = utils.load_model(mod.FourierGridModel, path)
    finally:
        torch.load = orig_load
    o, d, v = [torch.from_numpy(a) for a in synth.rays(41, 64, origin_scale=0.6)]
    o = o + torch.tensor([0.0, 1.0, -1.0])
    with torch.no_grad():
        a = mine(o, d, v, stepsize=0.5, render_depth=True)
        b = theirs(o, d, v, stepsize=0.5, render_depth=True)
    same = all(torch.equal(a[k], b[k]) for k in ("rgb_marched", "depth", "alphainv_last", "weights", "ray_id"))
    assert same, "reference model loaded from this package's checkpoint renders differently"
    res["reverse_checkpoint_ok"] = np.int64(1)
    np.savez_compressed(os.path.join(OUT, "train_utils.npz"), **res)
    print("train_utils", {k: v.tolist() for k, v in res.items() if k.endswith("_lr")}, "reverse ckpt ok")


def gen_dcvgo():
    """dcvgo.DirectContractedVoxGO.forward (contracted unbounded DVGOv2: cumdist_thres, mask cache, dense grids);
    num_voxels != num_voxels_base so that voxel_size_ratio != 1, a non-trivial mask, a scene cube off the origin."""
    dcvgo = install_stubs.import_reference("dcvgo")
    for name, seed, G, Gb, C, norm, R, dm, ds in synth.DCVGO_CASES:
        model = dcvgo.DirectContractedVoxGO(xyz_min=synth.DCVGO_BOX[0], xyz_max=synth.DCVGO_BOX[1], num_voxels=G ** 3,
                                            num_voxels_base=Gb ** 3, alpha_init=1e-2, fast_color_thres=1e-4,
                                            contracted_norm=norm, rgbnet_dim=C)
        ws = [int(x) for x in model.world_size]
        sd = model.state_dict()
        params = synth.dvgo_params(seed, ws, C, True, dens_mean=dm, dens_std=ds)
        with torch.no_grad():
            for k, v in params.items():
                assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
                sd[k].copy_(torch.from_numpy(v))
        model = model.to(DEVICE)
        o, d, v = [torch.from_numpy(a) for a in synth.rays(seed, R, origin_scale=0.5)]
        o = o + torch.tensor(synth.DCVGO_BOX[0]) * 0.5 + torch.tensor(synth.DCVGO_BOX[1]) * 0.5
        with torch.no_grad():
            out = model(o, d, v, stepsize=0.5, bg=1, render_depth=True)
        keep = {k: out[k].numpy() for k in ("alphainv_last", "weights", "wsum_mid", "rgb_marched", "raw_density", "raw_alpha",
                                            "raw_rgb", "ray_id", "step_id", "t", "s", "depth")}
        keep["n_max"] = np.int64(out["n_max"])
        keep["world_size"] = np.array(ws)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **keep)
        print(name, "world", ws, "M=%d of %d" % (out["weights"].numel(), R * out["n_max"]),
              "terminated %d/%d" % (int((out["alphainv_last"] < 1e-3).sum()), R), "wsum_mid mean %.3f" % float(out["wsum_mid"].mean()))


def gen_checkpoint():
    """A reference-format checkpoint (FourierGrid_ckpt_manager.py:44-51: model_kwargs + model_state_dict) of a tiny
    FourierGridModel with NON-trivial geometry (scene box != [-1,1]^3, num_voxels != num_voxels_base so that
    voxel_size_ratio != 1) plus the reference's render of a few rays: exercises
    FourierGridRenderer.from_reference_checkpoint."""
    mod = install_stubs.import_reference("FourierGrid_model")
    G, Gb, F, C = 9, 12, 2, 12
    model = mod.FourierGridModel(xyz_min=[-2.0, -1.0, -3.0], xyz_max=[2.0, 3.0, 1.0],
                                 num_voxels_density=G ** 3, num_voxels_base_density=Gb ** 3,
                                 num_voxels_rgb=G ** 3, num_voxels_base_rgb=Gb ** 3, num_voxels_viewdir=-1,
                                 alpha_init=1e-3, fast_color_thres=1e-4, fourier_freq_num=F, rgbnet_dim=C)
    Gd = int(model.world_len_density)
    params = synth.fouriergrid_params(41, Gd, F, C, dens_mean=4.0, dens_std=10.0)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(torch.from_numpy(v))
    ckpt = {'global_step': 123, 'model_kwargs': model.get_kwargs(), 'model_state_dict': model.state_dict()}
    torch.save(ckpt, os.path.join(OUT, "fg_ckpt_small.tar"))
    o, d, v = [torch.from_numpy(a) for a in synth.rays(41, 64, origin_scale=0.6)]
    o = o + torch.tensor([0.0, 1.0, -1.0])   # around the scene centre
    with torch.no_grad():
        out = model(o, d, v, stepsize=0.5, render_depth=True)
    np.savez_compressed(os.path.join(OUT, "fg_ckpt_small_render.npz"),
                        **{k: out[k].numpy() for k in ("rgb_marched", "depth", "alphainv_last")},
                        world_len=np.int64(Gd), interval=np.float32(float(0.5 * model.voxel_size_ratio_density)))
    print("checkpoint world_len", Gd, "ratio", float(model.voxel_size_ratio_density), "M", out["weights"].numel())


def gen_checkpoint_odd():
    """A reference checkpoint OUTSIDE the fused kernels' shapes (llff_default.py:31-32 / waymo_base.py:77 style): rgbnet
    width 64, depth 4, rgbnet_dim 9, viewbase_pe 8, colour grid at another resolution than the density grid -- rendered
    by the reference model; exercises fourier_render.ComposedFourierGridRenderer."""
    mod = install_stubs.import_reference("FourierGrid_model")
    model = mod.FourierGridModel(xyz_min=[-1.5, -1.0, -2.0], xyz_max=[1.5, 2.0, 1.0],
                                 num_voxels_density=10 ** 3, num_voxels_base_density=10 ** 3,
                                 num_voxels_rgb=7 ** 3, num_voxels_base_rgb=7 ** 3, num_voxels_viewdir=-1,
                                 alpha_init=1e-3, fast_color_thres=1e-4, fourier_freq_num=2, rgbnet_dim=9,
                                 rgbnet_width=64, rgbnet_depth=4, viewbase_pe=8)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        model.density.grid.copy_(torch.randn(model.density.grid.shape, generator=g) * 8.0 + 3.0)
        model.k0.grid.copy_(torch.randn(model.k0.grid.shape, generator=g) * 0.7)
        for p_ in model.rgbnet.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.25 if p_.dim() == 2 else 0.1))
    assert tuple(model.density.grid.shape[2:]) != tuple(model.k0.grid.shape[2:])
    ckpt = {'global_step': 7, 'model_kwargs': model.get_kwargs(), 'model_state_dict': model.state_dict()}
    torch.save(ckpt, os.path.join(OUT, "fg_ckpt_odd.tar"))
    o, d, v = [torch.from_numpy(a) for a in synth.rays(43, 96, origin_scale=0.5)]
    o = o + torch.tensor([0.0, 0.5, -0.5])
    with torch.no_grad():
        out = model(o, d, v, stepsize=0.5, render_depth=True)
    np.savez_compressed(os.path.join(OUT, "fg_ckpt_odd_render.npz"),
                        **{k: out[k].numpy() for k in ("rgb_marched", "depth", "alphainv_last")})
    print("odd checkpoint: density", tuple(model.density.grid.shape), "k0", tuple(model.k0.grid.shape), "M", out["weights"].numel())


def _voxgo_reference_model(kind, case):
    """the reference's own DirectVoxGO / DirectContractedVoxGO holding the synthetic parameters of a DVGO_CASES / DCVGO_CASES row"""
    if kind == "dvgo":
        dvgo = install_stubs.import_reference("dvgo")
        name, seed, G, C, direct, R, dm, ds = case
        xyz_min, xyz_max, nvox = dvgo_inputs(seed, G, C)
        model = dvgo.DirectVoxGO(xyz_min=xyz_min, xyz_max=xyz_max, num_voxels=nvox, num_voxels_base=nvox, alpha_init=1e-2,
                                 fast_color_thres=1e-4, rgbnet_dim=C, rgbnet_direct=direct, mask_cache_world_size=None)
    else:
        dcvgo = install_stubs.import_reference("dcvgo")
        name, seed, G, Gb, C, norm, R, dm, ds = case
        direct = True
        model = dcvgo.DirectContractedVoxGO(xyz_min=synth.DCVGO_BOX[0], xyz_max=synth.DCVGO_BOX[1], num_voxels=G ** 3,
                                            num_voxels_base=Gb ** 3, alpha_init=1e-2, fast_color_thres=1e-4,
                                            contracted_norm=norm, rgbnet_dim=C)
    ws = [int(x) for x in model.world_size]
    sd = model.state_dict()
    params = synth.dvgo_params(seed, ws, C, direct, dens_mean=dm, dens_std=ds)
    with torch.no_grad():
        for k, v in params.items():
            assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(v))
    return model.to(DEVICE), name, seed, R


def gen_voxgo_train():
    """One TRAINING forward + backward of the reference's own DirectVoxGO (all three DVGO_CASES: direct, residual colour, coarse)
    and DirectContractedVoxGO (both DCVGO_CASES) -- loss = mse + 0.01 * entropy_last (run_train.py:254-261), + the
    distortion-style use of weights / raw_density where the model returns them: outputs and the gradient of every parameter,
    to pin unboundednerfpytorch_amd.voxgo_model (fused training forward, tests/test_gpu_voxgo_train.py).  Also the models'
    parameter / buffer names and shapes, get_kwargs keys, update_occupancy_cache and scale_volume_grid."""
    for kind, cases in (("dvgo", DVGO_CASES), ("dcvgo", synth.DCVGO_CASES)):
        for case in cases:
            model, name, seed, R = _voxgo_reference_model(kind, case)
            if kind == "dvgo":
                o, d, v = [torch.from_numpy(a) for a in synth.rays(seed, R, origin_scale=0.4)]
                kw = dict(near=0.2, far=6.0, stepsize=0.5, bg=1, render_depth=True)
            else:
                o, d, v = [torch.from_numpy(a) for a in synth.rays(seed, R, origin_scale=0.5)]
                o = o + torch.tensor(synth.DCVGO_BOX[0]) * 0.5 + torch.tensor(synth.DCVGO_BOX[1]) * 0.5
                kw = dict(stepsize=0.5, bg=1, render_depth=True)
            target = torch.from_numpy(synth.uniform(seed + 5, R * 3).reshape(R, 3))
            out = model(o, d, v, global_step=1, is_train=True, **kw)
            loss = torch.nn.functional.mse_loss(out["rgb_marched"], target)
            pout = out["alphainv_last"].clamp(1e-6, 1 - 1e-6)
            loss = loss + 0.01 * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout))).mean()
            loss = loss + 0.05 * (out["weights"] * out["weights"]).sum() / R
            if "raw_density" in out:
                loss = loss + 1e-4 * out["raw_density"].sum() / R
            loss.backward()
            res = {"loss": loss.detach().numpy(), "n_kept": np.int64(out["weights"].numel()), "target": target.numpy()}
            for k in ("rgb_marched", "alphainv_last", "weights", "ray_id", "raw_rgb", "raw_alpha", "depth", "step_id", "wsum_mid"):
                if k in out:
                    res[k] = out[k].detach().numpy()
            for k, p in model.named_parameters():
                if p.grad is not None:
                    res["grad." + k] = p.grad.numpy()
            sd = model.state_dict()
            res["sd_keys"] = np.array(sorted(sd.keys()))
            res["sd_shapes"] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])
            res["kwargs_keys"] = np.array(sorted(model.get_kwargs().keys()))
            model.zero_grad()
            model.update_occupancy_cache()
            res["occ_mask"] = model.mask_cache.mask.numpy().copy()
            G2 = int(model.world_size[0]) + 5
            model.scale_volume_grid(G2 ** 3)
            res["scaled_density"] = model.density.grid.detach().numpy().copy()
            res["scaled_k0"] = model.k0.grid.detach().numpy().copy()
            res["scaled_mask"] = model.mask_cache.mask.numpy().copy()
            res["scaled_world_size"] = model.world_size.numpy().copy()
            res["scaled_ratio"] = np.float32(float(model.voxel_size_ratio))
            res["scaled_num_voxels"] = np.int64(G2 ** 3)
            with torch.no_grad():
                out2 = model(o, d, v, global_step=2, is_train=True, **kw)
            res["scaled_rgb_marched"] = out2["rgb_marched"].numpy()
            res["scaled_n_kept"] = np.int64(out2["weights"].numel())
            np.savez_compressed(os.path.join(OUT, "voxgo_train_" + name + ".npz"), **res)
            print("voxgo_train", name, "loss %.6f kept %d, after scaling to %s kept %d" % (
                float(loss), int(res["n_kept"]), res["scaled_world_size"].tolist(), int(res["scaled_n_kept"])))


def gen_fourier_loss():
    """The image-space Fourier loss (`weight_freq`; bicycle_single.py:57 and stump_single.py:55 use 5.0):
    (a) the reference's own FourierMSELoss module (FourierGrid_model.py:112-129) on seeded [R,3] colours: value and gradient;
    (b) one training forward + backward of the reference FourierGridModel on synth.TRAIN_CASE with the loss run_train.py:254-265 forms
        under bicycle_single.py's weights that need no third-party package (weight_main 1.0, weight_freq 5.0, weight_entropy_last 0.001,
        weight_nearclip 1.0 with near_thres = FREQ_NEAR; weight_distortion calls torch_efficient_distloss, absent from the image):
        loss, mse, freq term and the gradient of every parameter."""
    mod = install_stubs.import_reference("FourierGrid_model")
    crit = mod.FourierMSELoss()
    R = 257
    pred = torch.from_numpy(synth.uniform(901, R * 3).reshape(R, 3)).requires_grad_(True)
    gt = torch.from_numpy(synth.uniform(902, R * 3).reshape(R, 3))
    val = crit(pred, gt)
    val.backward()
    res = {"a_loss": val.detach().numpy(), "a_grad": pred.grad.numpy().copy()}
    c = synth.TRAIN_CASE
    params = synth.fouriergrid_params(c["seed"], c["G"], c["F"], c["C"], viewbase_pe=c["pe"], dens_mean=c["dm"], dens_std=c["ds"])
    model = build_reference_model(mod, c["G"], c["F"], c["C"], c["pe"], c["norm"], c["thres"], params)
    o, d, v = [torch.from_numpy(a) for a in synth.rays(c["seed"], c["R"])]
    target = torch.from_numpy(synth.uniform(c["seed"] + 5, c["R"] * 3).reshape(c["R"], 3))
    out = model(o, d, v, global_step=1, is_train=True, stepsize=c["stepsize"], render_depth=True)
    w = synth.FREQ_WEIGHTS
    mse = torch.nn.functional.mse_loss(out["rgb_marched"], target)                       # run_train.py:254
    freq = crit(out["rgb_marched"], target)                                              # :255
    loss = w["weight_main"] * mse + w["weight_freq"] * freq                              # :257
    pout = out["alphainv_last"].clamp(1e-6, 1 - 1e-6)                                    # :258-261
    loss = loss + w["weight_entropy_last"] * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout))).mean()
    near_mask = out["t"] < synth.FREQ_NEAR                                               # :262-268
    density = out["raw_density"][near_mask]
    assert len(density) > 0
    loss = loss + w["weight_nearclip"] * (density - density.detach()).sum()
    loss.backward()
    res.update({"b_loss": loss.detach().numpy(), "b_mse": mse.detach().numpy(), "b_freq": freq.detach().numpy(),
                "b_n_kept": np.int64(out["weights"].numel()), "b_n_near": np.int64(int(near_mask.sum())),
                "b_rgb_marched": out["rgb_marched"].detach().numpy()})
    for k, p in model.named_parameters():
        if p.grad is not None:
            res["b_grad." + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "fourier_loss.npz"), **res)
    print("fourier_loss (a) %.6f (b) loss %.6f mse %.6f freq %.6f kept %d near %d" % (
        float(val), float(loss), float(mse), float(freq), int(res["b_n_kept"]), int(res["b_n_near"])))


if __name__ == "__main__":
    torch.set_num_threads(1)  # deterministic reduction order in F.linear / grid_sample
    gen_fouriergrid()
    gen_grid_query()
    gen_autograd_and_adam()
    gen_rays_view()
    gen_dvgo()
    gen_checkpoint()
    gen_checkpoint_odd()
    gen_distortion()
    gen_train_step()
    gen_dcvgo()
    gen_dvgo_utils()
    gen_model_utils()
    gen_train_utils()
    gen_voxgo_train()
    gen_fourier_loss()
