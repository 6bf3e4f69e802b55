"""Config-3 shaped parity (BASELINE.json configs[2]): one reference-style train step -- forward with autograd
through Raw2Alpha / Alphas2Weights, backward, total-variation gradient, MaskedAdam -- run on CPU with the oracle
ops and on the GPU with the product's drop-in ops, then the updated parameters are compared.

torch's grid_sample / Linear run on different back-ends on the two sides (CPU vs ROCm, atomics in the
grid_sample backward), so gradients agree to ~1e-5 relative, not bit for bit; Adam's first steps move every
touched voxel by ~lr, which makes the parameter comparison sensitive to WHICH voxels are touched (exact-zero
gradient mask) rather than to gradient noise.
"""
import numpy as np
import pytest
import torch

import synth
from oracle import model_oracle, ref_ops
from test_oracle_golden import make_state

pytestmark = pytest.mark.gpu


def run_steps(device, backend_ru, backend_tv, adam_cls, n_steps, seed=21, grid_query=None):
    G, F, C, R, stepsize = 20, 3, 12, 512, 0.5
    cfg = make_state(seed, G, F, C, 4, "inf", 1e-4, 4.0, 10.0)
    params = {
        'density_grid': cfg['density_grid'].clone(), 'k0_grid': cfg['k0_grid'].clone(),
        'w0': cfg['rgbnet_weights'][0].clone(), 'b0': cfg['rgbnet_biases'][0].clone(),
        'w1': cfg['rgbnet_weights'][1].clone(), 'b1': cfg['rgbnet_biases'][1].clone(),
        'w2': cfg['rgbnet_weights'][2].clone(), 'b2': cfg['rgbnet_biases'][2].clone(),
    }
    params = {k: torch.nn.Parameter(v.to(device)) for k, v in params.items()}
    opt = adam_cls([
        {'params': [params['density_grid']], 'lr': 0.1, 'skip_zero_grad': True},
        {'params': [params['k0_grid']], 'lr': 0.1, 'skip_zero_grad': True},
        {'params': [params[k] for k in ('w0', 'b0', 'w1', 'b1', 'w2', 'b2')], 'lr': 1e-3, 'skip_zero_grad': False}])
    Raw2Alpha, Alphas2Weights = model_oracle.make_autograd_ops(backend_ru)
    o, d, v = [torch.from_numpy(a).to(device) for a in synth.rays(seed, R)]
    target = torch.from_numpy(synth.uniform(seed + 5, R * 3).reshape(R, 3)).to(device)
    info = []
    for step in range(n_steps):
        opt.zero_grad(set_to_none=True)
        out = model_oracle.fouriergrid_train_forward(params, cfg, o, d, v, stepsize, Raw2Alpha, Alphas2Weights,
                                                     grid_query=grid_query)
        loss = torch.nn.functional.mse_loss(out['rgb_marched'], target)
        pout = out['alphainv_last'].clamp(1e-6, 1 - 1e-6)
        loss = loss + 0.01 * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout))).mean()  # entropy_last
        loss.backward()
        # FourierGrid_model.py:482-488 / run_train.py:281-287: w = weight * world_size.max() / 128, dense for the first step
        w = 1e-5 * G / 128
        backend_tv.total_variation_add_grad(params['density_grid'], params['density_grid'].grad, w, w, w, step == 0)
        backend_tv.total_variation_add_grad(params['k0_grid'], params['k0_grid'].grad, w, w, w, step == 0)
        info.append((float(loss.detach()), out['n_kept'], {k: p.grad.detach().cpu().clone() for k, p in params.items()}))
        opt.step()
    return {k: p.detach().cpu() for k, p in params.items()}, info


class _OracleAdam(torch.optim.Optimizer):
    """MaskedAdam dispatch logic (masked_adam.py:43-75) over the oracle kernels, for the CPU side."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                fn = ref_ops.masked_adam_upd if g['skip_zero_grad'] else ref_ops.adam_upd
                fn(p, p.grad, st['exp_avg'], st['exp_avg_sq'], st['step'], g['betas'][0], g['betas'][1], g['lr'], g['eps'])


def test_train_step_gpu_matches_cpu_oracle():
    from unboundednerfpytorch_amd import render_utils_cuda, total_variation_cuda
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    torch.set_num_threads(8)
    n_steps = 2
    p_cpu, info_cpu = run_steps("cpu", ref_ops, ref_ops, _OracleAdam, n_steps)
    # GPU side: every grid lookup (forward AND backward) on the HIP kernels too -- the whole step then runs on the
    # product's ops except the three rgbnet Linear layers (plain rocBLAS GEMMs through torch)
    from unboundednerfpytorch_amd.grid import GridQuery
    p_gpu, info_gpu = run_steps("cuda", render_utils_cuda, total_variation_cuda, MaskedAdam, n_steps,
                                grid_query=GridQuery.apply)
    for (l0, n0, g0), (l1, n1, g1) in zip(info_cpu, info_gpu):
        assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)), (l0, l1)
        assert abs(n0 - n1) <= 2
    # gradients of the first step (same parameters on both sides)
    g0, g1 = info_cpu[0][2], info_gpu[0][2]
    for k in g0:
        scale = float(g0[k].abs().max()) + 1e-12
        err = float((g0[k] - g1[k]).abs().max()) / scale
        assert err < 5e-4, (k, err)
    # the touched-voxel masks of the sparse grid gradients must agree (dense TV in step 0 touches everything,
    # so look at step 1)
    for k in ('density_grid', 'k0_grid'):
        m0, m1 = info_cpu[1][2][k] != 0, info_gpu[1][2][k] != 0
        assert float((m0 != m1).float().mean()) < 1e-4, k
    for k in p_cpu:
        diff = (p_cpu[k] - p_gpu[k]).abs()
        # Adam moves a touched entry by <= lr per step; disagreement beyond a few % of that is a real mismatch
        lr = 0.1 if 'grid' in k else 1e-3
        frac_bad = float((diff > 0.05 * lr).float().mean())
        assert frac_bad < 2e-3, (k, frac_bad, float(diff.max()))
