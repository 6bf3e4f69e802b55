"""One iteration of the reference's training loop (run_train.py:185-296) on this package's model and optimizers:
progressive grid scaling at the `pg_scale` steps, forward, the loss terms the train config switches on, backward, the
total-variation gradient inside its step window, the (masked) Adam step and the continuous learning-rate decay.
Data loading, ray-batch sampling, logging and checkpoint scheduling stay with the caller (the reference's
`scene_rep_reconstruction` does them inline around these lines)."""
import torch
import torch.nn.functional as F

from . import _gradpool
from .train_utils import create_optimizer_or_freeze_model


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def fourier_mse_loss(pred, gt):
    """MSE between the real parts of the FFTs over the colour axis (FourierMSELoss, FourierGrid_model.py:115-133)."""
    return F.mse_loss(torch.fft.fft(pred, dim=-1).real, torch.fft.fft(gt, dim=-1).real)


def maybe_scale_grids(model, optimizer, cfg_train, cfg_model, global_step, **optimizer_kw):
    """run_train.py:186-201: at a `pg_scale` step the grids grow to num_voxels / 2^(scales still to come), the
    optimizer is rebuilt (its state refers to the old grids) and the density bias is lowered by `decay_after_scale`."""
    pg = list(_get(cfg_train, 'pg_scale', []))
    if global_step not in pg:
        return optimizer
    rest = len(pg) - pg.index(global_step) - 1
    if hasattr(model, 'num_voxels_density'):
        model.scale_volume_grid(int(_get(cfg_model, 'num_voxels_density') / (2 ** rest)),
                                int(_get(cfg_model, 'num_voxels_rgb') / (2 ** rest)))
    else:       # voxgo_model.DirectVoxGO / DirectContractedVoxGO: one resolution for both grids.  The reference's configs carry it
        # as `num_voxels_rgb` (run_train.py:190-194 scales these models with cur_voxels_rgb); a plain `num_voxels` is accepted too
        nv = _get(cfg_model, 'num_voxels', None)
        if nv is None:
            nv = _get(cfg_model, 'num_voxels_rgb')
        model.scale_volume_grid(int(nv / (2 ** rest)))
    optimizer = create_optimizer_or_freeze_model(model, cfg_train, global_step=0, **optimizer_kw)
    model.act_shift -= _get(cfg_train, 'decay_after_scale', 0.0)
    return optimizer


def training_loss(render_result, target, cfg_train, n_rays, near_thres=None, distortion_fn=None, world_size=1):
    """The loss of run_train.py:254-279 from the model's return dict.  near_thres = near_clip / scene_radius[0] when
    weight_nearclip is used; distortion_fn(w, s, interval, ray_id) defaults to ops.flatten_eff_distloss (the library call
    of run_train.py:274, gradient = derivative of the value, 1/n_rays included).
    world_size > 1 (data-parallel, gradients AVERAGED over ranks by ShardedMaskedAdam): every term that is a mean over
    the local rays needs no change; the one SUM-type term (nearclip: +1 per near sample) is multiplied by world_size
    so that the rank average equals the whole-batch sum."""
    mse = F.mse_loss(render_result['rgb_marched'], target)
    loss = _get(cfg_train, 'weight_main', 1.0) * mse
    w_freq = _get(cfg_train, 'weight_freq', 0.0)
    if w_freq:
        loss = loss + w_freq * fourier_mse_loss(render_result['rgb_marched'], target)
    if _get(cfg_train, 'weight_entropy_last', 0.0) > 0:
        p = render_result['alphainv_last'].clamp(1e-6, 1 - 1e-6)
        loss = loss + _get(cfg_train, 'weight_entropy_last') * (-(p * torch.log(p) + (1 - p) * torch.log(1 - p))).mean()
    if _get(cfg_train, 'weight_nearclip', 0.0) > 0:
        # run_train.py:262-265 selects the near samples with a boolean index and skips the term when there are none; the
        # value of the term is 0 either way and its gradient is +weight on the selected densities -- a masked sum gives
        # both without the nonzero + host sync
        dens = render_result['raw_density']
        near = (render_result['t'] < near_thres).to(dens.dtype)
        loss = loss + (_get(cfg_train, 'weight_nearclip') * world_size) * ((dens - dens.detach()) * near).sum()
    if _get(cfg_train, 'weight_distortion', 0.0) > 0 and render_result['weights'].numel() > 0:
        if distortion_fn is None:
            from .ops import flatten_eff_distloss as distortion_fn
        loss = loss + _get(cfg_train, 'weight_distortion') * distortion_fn(
            render_result['weights'], render_result['s'], 1 / render_result['n_max'], render_result['ray_id'])
    if _get(cfg_train, 'weight_rgbper', 0.0) > 0:
        per = (render_result['raw_rgb'] - target[render_result['ray_id']]).pow(2).sum(-1)
        loss = loss + _get(cfg_train, 'weight_rgbper') * (per * render_result['weights'].detach()).sum() / n_rays
    return loss, mse


def _mark(timers, name):
    """timers: optional dict name -> list of torch.cuda.Event pairs (tools/bench_train_step.py splits a step by phase)"""
    if timers is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        timers.setdefault(name, []).append(ev)


def train_iteration(model, optimizer, rays_o, rays_d, viewdirs, target, cfg_train, global_step, render_kwargs,
                    near_thres=None, distortion_fn=None, decay_lr=True, world_size=1, timers=None, overlap_k0_update=False,
                    return_tensors=False):
    """Forward ... optimizer.step() of one global_step (call maybe_scale_grids first).  Returns (loss, psnr).
    Data-parallel use (ShardedMaskedAdam averages the ranks' gradients): pass world_size so that the total-variation
    term, which the reference scales by 1 / len(rays_o), is scaled by the GLOBAL batch size and the sum-type nearclip
    term by world_size -- the ray-mean losses (mse, entropy, distortion, rgbper) need no change (mean over the local
    rays, then mean over ranks; exact when every rank's last ray has a sample, since the distortion loss normalises by
    ray_id.max()+1 like the library).  TV itself runs inside optimizer.step on the
    REDUCED gradient (grad_hook), so its masked mode sees the voxels any rank touched: the data-parallel step equals
    the single-process step on the whole batch in both TV phases (tests/test_host_logic.py).
    overlap_k0_update (single process, HIP optimizer): see ShardedMaskedAdam.step(overlap=...) -- same results; k0.grid
    must then be read through the model (forward, state_dict, ...) or after torch.cuda.synchronize().
    return_tensors: return (loss, psnr) as 0-d device tensors instead of Python floats -- no host sync at the end of the
    iteration (the reference reads psnr.item() every step, run_train.py:297; a caller that logs every N steps need not)."""
    _mark(timers, "start")
    n_rays = len(rays_o)
    kw = render_kwargs
    if distortion_fn is None and getattr(model, 'fused_loss', False) and rays_o.is_cuda:
        # the HIP model composites and evaluates the loss in one op (ops.RenderLoss); n_max is known before the forward
        from .ops import loss_coefficients
        coef = loss_coefficients(cfg_train, n_rays, model.sample_table(render_kwargs['stepsize'], rays_o.device).numel(),
                                 near_thres, world_size)
        if coef is not None:
            kw = dict(render_kwargs, fused_loss={'target': target, 'coef': coef})
    out = model(rays_o, rays_d, viewdirs, global_step=global_step, is_train=True, **kw)
    _mark(timers, "forward")
    optimizer.zero_grad(set_to_none=True)
    if 'loss' in out:
        loss, mse = out['loss'], out['mse']
    else:
        loss, mse = training_loss(out, target, cfg_train, n_rays, near_thres, distortion_fn, world_size)
    _mark(timers, "loss")
    tv_on = (global_step < _get(cfg_train, 'tv_before', 0) and global_step > _get(cfg_train, 'tv_after', 0)
             and global_step % _get(cfg_train, 'tv_every', 1) == 0)
    tv_terms = None
    if tv_on:
        # run_train.py:281-287.  The TV term is applied to the gradient the optimizer is about to use -- inside
        # optimizer.step, i.e. AFTER the cross-rank reduction in data-parallel runs (in a single process that is exactly
        # "TV, then step", and in dense mode the HIP optimizer fuses the two passes).  weight / batch size, then the
        # models' own scaling by world_size.max() / 128.
        dense = global_step < _get(cfg_train, 'tv_dense_before', 0)
        n_global = n_rays * world_size
        tv_terms = {}
        if _get(cfg_train, 'weight_tv_density', 0.0) > 0:
            tv_terms[model.density.grid] = (float(_get(cfg_train, 'weight_tv_density') / n_global * model.world_size_density.max() / 128),
                                            dense, model.density.tv_module)
        if _get(cfg_train, 'weight_tv_k0', 0.0) > 0:
            tv_terms[model.k0.grid] = (float(_get(cfg_train, 'weight_tv_k0') / n_global * model.world_size_rgb.max() / 128),
                                       dense, model.k0.tv_module)
    overlap = bool(tv_terms and overlap_k0_update and world_size == 1 and model.k0.grid in tv_terms)
    hook = None
    if overlap and hasattr(optimizer, 'step_param') and hasattr(model.k0.grid, 'register_post_accumulate_grad_hook'):
        # the k0 update (the step's largest pass, HBM-bound) is queued on the second stream THE MOMENT the k0 gradient is
        # complete -- the k0 lookup's backward runs before the density path's in the autograd order -- and so runs beside the
        # density backward, the density / rgbnet updates, the next iteration's march, host syncs and launch-bound glue
        k0_term = tv_terms[model.k0.grid]

        def _early_k0_update(p):          # (an autograd hook must return None)
            optimizer.step_param(p, k0_term, overlap=True)
        if out.get('native') is not None:
            # the native step is ONE autograd node: its backward calls back between its two halves (native_step.VoxGOStep)
            out['native']['k0_grad_ready'] = _early_k0_update
        else:
            hook = model.k0.grid.register_post_accumulate_grad_hook(_early_k0_update)
    # Touched-line bitmaps (_gradpool): in THIS loss graph the feature grid receives gradient from exactly one lookup (the
    # model's k0 query; every loss term reaches the grid through it), which is what the bitmap's validity rests on -- so the
    # step certifies it for its own backward only.  A caller-supplied loss that regularises k0.grid directly must not use
    # train_iteration's certificate: pass distortion_fn / losses through `out`, or call _gradpool.decertify first.
    # (data-parallel runs use the bitmap to pick the lines they exchange, sharded_adam._line_bits)
    certified = [model.k0.grid] if (hasattr(model, 'k0') and isinstance(getattr(model.k0, 'grid', None), torch.nn.Parameter)) else []
    _gradpool.certify(certified)
    try:
        try:
            loss.backward()
        except BaseException:
            early = getattr(optimizer, '_early', None)     # a hook may have stepped k0 before the backward failed: the next
            if early:                                       # step() must not skip a parameter on the strength of this one
                early.clear()
            raise
        finally:
            if hook is not None:
                hook.remove()
        _mark(timers, "backward")
        if overlap:
            optimizer.step(tv_terms=tv_terms, overlap=[model.k0.grid])     # (k0 is skipped here when the hook has updated it)
        elif tv_terms:
            optimizer.step(tv_terms=tv_terms)
        else:
            optimizer.step()
    finally:
        _gradpool.decertify(certified)
    _mark(timers, "tv+adam")
    if decay_lr:                      # run_train.py:290-295 (the reference skips this for FourierGrid on tankstemple)
        factor = 0.1 ** (1 / (_get(cfg_train, 'lrate_decay') * 1000))
        for g in optimizer.param_groups:
            g['lr'] = g['lr'] * factor
    if not return_tensors and out.get('loss_mse') is not None:
        # the native step hands out {loss, mse} as one [2] tensor: one device-to-host copy for both, the psnr formed on the host in
        # float32 like the tensor expression below (no log10 / scale launches, no second synchronising read)
        import numpy as np
        loss_v, mse_v = out['loss_mse'].tolist()
        return loss_v, float(np.float32(-10.0) * np.log10(np.float32(mse_v)))
    psnr = -10.0 * torch.log10(mse.detach())
    if return_tensors:
        return loss.detach(), psnr
    return float(loss.detach()), float(psnr)
