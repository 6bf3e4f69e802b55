"""oracle/ref_train.py -- TEST INFRASTRUCTURE ONLY (tests/test_gpu_train_long.py, tools/).

The reference's own training loop, driven on THIS GPU: its own `FourierGridModel` (FourierGrid/FourierGrid_model.py), its own
`utils.create_optimizer_or_freeze_model` and `MaskedAdam` (utils.py:26-56, masked_adam.py), its own total-variation methods,
over its own compiled kernels (oracle/_ref/<variant>, backend "kernels:fma") or the C restatement (backend "oracle", CPU).

`scene_rep_reconstruction` (run_train.py:62-330) is one function that also loads data, parses the mmcv config and writes
checkpoints; what it does per iteration is lines 186-296.  `run()` below drives the reference's objects through exactly those
steps, citing the line each one restates:

    :187-201  progressive grid scaling at a pg_scale step: model.scale_volume_grid, a NEW optimizer, act_shift -= decay_after_scale
    :246-248  forward(rays_o, rays_d, viewdirs, global_step, is_train=True, **render_kwargs); optimizer.zero_grad(set_to_none=True)
    :249-254  mse, psnr, weight_main * mse (+ weight_freq * FourierMSELoss when the config asks for it)
    :255-258  entropy of alphainv_last      :259-265  nearclip      :273-276  per-point rgb loss
    :277-286  TV gradients inside the (tv_after, tv_before) window, dense while global_step < tv_dense_before
    :287      optimizer.step()               :290-295  continuous lr decay
(weight_distortion is not driven: run_train.py:266-272 calls the third-party torch_efficient_distloss, absent from the image.)
Ray batches come from the caller (the reference draws them with its own sampler, :203-236; both sides of a comparison must see the
same rays).  Nothing here is imported by the product."""
import importlib
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


def import_reference(backend):
    """(FourierGrid_model module, utils module) of the reference with `backend`'s extension modules bound"""
    from oracle import ref_model
    mod = ref_model._import_reference(backend)
    utils = importlib.import_module("FourierGrid.utils")
    return mod, utils


def build_model(backend, ctor_kwargs, device, kind="fourier"):
    """kind: "fourier" = FourierGrid_model.FourierGridModel, "dvgo" = dvgo.DirectVoxGO, "dcvgo" = dcvgo.DirectContractedVoxGO"""
    mod, _ = import_reference(backend)
    with torch.device("cpu"):
        if kind == "fourier":
            model = mod.FourierGridModel(**ctor_kwargs)
        elif kind == "dvgo":
            model = importlib.import_module("FourierGrid.dvgo").DirectVoxGO(**ctor_kwargs)
        elif kind == "dcvgo":
            model = importlib.import_module("FourierGrid.dcvgo").DirectContractedVoxGO(**ctor_kwargs)
        else:
            raise ValueError(kind)
    return model.to(device)


def run(model, backend, cfg_train, cfg_model, batches, n_iters, device, eval_fn=None, eval_every=20, near_thres=None, render_kwargs=None,
        on_step=None):
    """Train the reference `model` for n_iters iterations on `batches[i] = (rays_o, rays_d, viewdirs, target)` (device tensors).
    Returns {"psnr_train": [...], "loss": [...], "eval": [(step, value of eval_fn(model))...]}.  cfg_train / cfg_model: dicts."""
    _, utils = import_reference(backend)
    ct = SimpleNamespace(**cfg_train)
    ct.keys = lambda: cfg_train.keys()          # utils.create_optimizer_or_freeze_model iterates cfg_train.keys() (an mmcv Config there)
    rk = dict(render_kwargs or {})
    dev = torch.device(device)
    hist = {"psnr_train": [], "loss": [], "eval": []}
    # The reference program makes CUDA the default tensor type (run_FourierGrid.py: torch.set_default_tensor_type(
    # 'torch.cuda.FloatTensor')) and its model code relies on it: the sample table (FourierGrid_model.py:526-532) and the
    # legacy torch.Tensor(list) constructors of MaskGrid (FourierGrid_grid.py:156-157, reached from scale_volume_grid) carry no
    # device.  Same here for the duration of the run; the caller's default is restored afterwards.
    legacy = dev.type == "cuda"
    if legacy:
        torch.set_default_tensor_type(torch.cuda.FloatTensor)
    try:
        with torch.device(dev):
            _loop(model, utils, ct, cfg_train, cfg_model, batches, n_iters, rk, near_thres, eval_fn, eval_every, on_step, hist)
    finally:
        if legacy:
            torch.set_default_tensor_type(torch.FloatTensor)
    return hist


def _loop(model, utils, ct, cfg_train, cfg_model, batches, n_iters, rk, near_thres, eval_fn, eval_every, on_step, hist):
    if True:
        optimizer = utils.create_optimizer_or_freeze_model(model, ct, global_step=0)
        pg = list(cfg_train.get("pg_scale", []))
        for global_step in range(1, n_iters + 1):
            if global_step in pg:                                                                   # run_train.py:187-201
                n_rest = len(pg) - pg.index(global_step) - 1
                if "num_voxels_density" in cfg_model:
                    model.scale_volume_grid(int(cfg_model["num_voxels_density"] / (2 ** n_rest)), int(cfg_model["num_voxels_rgb"] / (2 ** n_rest)))
                else:       # DirectVoxGO / DirectContractedVoxGO: one resolution (run_train.py:190-196)
                    model.scale_volume_grid(int(cfg_model["num_voxels"] / (2 ** n_rest)))
                optimizer = utils.create_optimizer_or_freeze_model(model, ct, global_step=0)
                model.act_shift -= cfg_train.get("decay_after_scale", 0.0)
            rays_o, rays_d, viewdirs, target = batches[global_step - 1]
            out = model(rays_o, rays_d, viewdirs, global_step=global_step, is_train=True, **rk)     # :246-247
            optimizer.zero_grad(set_to_none=True)                                                   # :248
            mse = F.mse_loss(out["rgb_marched"], target)                                            # :249
            psnr = -10.0 * torch.log10(mse.detach())                                                # :251 (utils.mse2psnr)
            loss = cfg_train["weight_main"] * mse                                                   # :252
            if cfg_train.get("weight_entropy_last", 0) > 0:                                         # :255-258
                pout = out["alphainv_last"].clamp(1e-6, 1 - 1e-6)
                loss = loss + cfg_train["weight_entropy_last"] * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean())
            if cfg_train.get("weight_nearclip", 0) > 0:                                             # :259-265
                near_mask = out["t"] < near_thres
                density = out["raw_density"][near_mask]
                if len(density):
                    loss = loss + cfg_train["weight_nearclip"] * (density - density.detach()).sum()
            assert not cfg_train.get("weight_distortion", 0), "third-party torch_efficient_distloss is not available: set weight_distortion = 0"
            if cfg_train.get("weight_rgbper", 0) > 0:                                               # :273-276
                rgbper = (out["raw_rgb"] - target[out["ray_id"]]).pow(2).sum(-1)
                loss = loss + cfg_train["weight_rgbper"] * (rgbper * out["weights"].detach()).sum() / len(rays_o)
            loss.backward()                                                                         # :277
            if global_step < cfg_train["tv_before"] and global_step > cfg_train["tv_after"] and global_step % cfg_train["tv_every"] == 0:   # :278-286
                dense = global_step < cfg_train["tv_dense_before"]
                if cfg_train.get("weight_tv_density", 0) > 0:
                    model.density_total_variation_add_grad(cfg_train["weight_tv_density"] / len(rays_o), dense)
                if cfg_train.get("weight_tv_k0", 0) > 0:
                    model.k0_total_variation_add_grad(cfg_train["weight_tv_k0"] / len(rays_o), dense)
            optimizer.step()                                                                        # :287
            decay_factor = 0.1 ** (1 / (cfg_train["lrate_decay"] * 1000))                           # :290-295
            for g in optimizer.param_groups:
                g["lr"] = g["lr"] * decay_factor
            hist["psnr_train"].append(float(psnr))
            hist["loss"].append(float(loss.detach()))
            if on_step is not None:
                on_step(global_step, model)
            if eval_fn is not None and (global_step % eval_every == 0 or global_step == n_iters):
                hist["eval"].append((global_step, eval_fn(model)))

