"""oracle/model_oracle.py -- TEST INFRASTRUCTURE ONLY.

Torch-CPU restatement of the *Python* half of the reference hot path: the parts the reference
itself expresses with third-party torch ops (F.grid_sample, nn.Linear, torch_scatter.segment_coo)
around its native kernels.  Native-kernel arithmetic comes from the C oracle (oracle/ref_ops.py).

  fourier_grid_query      <- FourierGrid/FourierGrid_grid.py:21-36,60-78 (+ grid.py:50-61, P=1)
  contracted_sample_ray   <- FourierGrid/FourierGrid_model.py:509-552  (dcvgo.py:228-262 for t_boundary=2)
  rgbnet / viewdir emb    <- FourierGrid/FourierGrid_model.py:231-241,631-637
  fouriergrid_render      <- FourierGrid/FourierGrid_model.py:554-672 (forward)

PARITY STATUS: pinned.  tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tests/golden/gen_golden.py produced by running the reference's own
FourierGrid_model.py (imported from /root/reference) in the build container.

Third-party arithmetic restated through the same library calls the reference makes:
torch.nn.functional.grid_sample(mode='bilinear', align_corners=True, zero padding) -- torch is not
pinned by the reference (requirements.txt) beyond README "pytorch==1.13.1"; here torch 2.10 CPU.
"""
import math

import torch
import torch.nn.functional as F

from oracle import ref_ops


# How sin / cos of the Fourier levels are evaluated.  "torch" (default) = torch.sin / torch.cos on fp32 tensors, i.e. what
# the reference's Python does on this host (Sleef's 1-ulp vector kernels).  "rounded64" = the correctly rounded fp32
# results (evaluated in fp64, rounded once): a second, equally conforming libm.  The two differ by 1 ulp on a fraction
# of the arguments; tests use the pair to measure how far two VALID evaluations of the reference formula are apart on
# a given scene (tests/test_gpu_s1_scale.py) -- the arithmetic of everything else is identical.
PE_MATH = "torch"


def pe_levels(u, freq_num):
    """[..., 3] normalised coords -> list of P=1+2F coordinate triples
    [u, sin(1u), cos(1u), sin(2u), cos(2u), ...]  (FourierGrid_grid.py:26-36,70; no pi)."""
    levels = [u]
    for k in range(freq_num):
        f = float(2 ** k)
        if PE_MATH == "rounded64":
            x = (f * u).double()
            levels.append(torch.sin(x).float())
            levels.append(torch.cos(x).float())
        else:
            levels.append(torch.sin(f * u))
            levels.append(torch.cos(f * u))
    return levels


def fourier_grid_query(grid, xyz, xyz_min, xyz_max, freq_num):
    """grid [P,C,X,Y,Z] (P=1+2F, or P=1 with freq_num<=0 for a DenseGrid); xyz [...,3] world
    coordinates.  Returns [...,C] (squeezed when C==1): mean over levels of trilinear taps."""
    lead = xyz.shape[:-1]
    C = grid.shape[1]
    pts = xyz.reshape(-1, 3)
    # world (x,y,z) -> grid_sample order (z,y,x) so that x indexes dim 2 of the grid
    u = ((pts - xyz_min) / (xyz_max - xyz_min)).flip((-1,)) * 2 - 1
    if freq_num > 0:
        coords = torch.stack(pe_levels(u, freq_num), 0)          # [P,n,3]
        assert coords.shape[0] == grid.shape[0]
        taps = F.grid_sample(grid, coords[:, None, None], mode='bilinear', align_corners=True)
        val = taps.mean(0)                                        # [C,1,1,n]
    else:
        assert grid.shape[0] == 1
        val = F.grid_sample(grid, u[None, None, None], mode='bilinear', align_corners=True)[0]
    out = val.reshape(C, -1).T.reshape(*lead, C)
    return out.squeeze(-1) if C == 1 else out


def sample_t(world_len, stepsize, bg_len, t_boundary=1.5):
    """Mid-point sample distances shared by every ray (FourierGrid_model.py:524-532)."""
    n_inner = int(2 / (2 + 2 * bg_len) * world_len / stepsize) + 1
    b_in = torch.linspace(0, t_boundary, n_inner + 1)
    b_out = t_boundary / torch.linspace(1, 1 / 128, n_inner + 1)
    return torch.cat([(b_in[1:] + b_in[:-1]) * 0.5, (b_out[1:] + b_out[:-1]) * 0.5])


def contracted_sample_ray(rays_o, rays_d, scene_center, scene_radius, t, bg_len, contracted_norm='inf'):
    """-> ray_pts [R,S,3], inner_mask [R,S].  Contraction p/|p| * ((1+bg) - bg/|p|) outside the
    unit cube/ball (FourierGrid_model.py:521-552 with order=1, seperate_boundary=1)."""
    o = (rays_o - scene_center) / scene_radius
    d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    pts = o[:, None, :] + d[:, None, :] * t[None, :, None]
    if contracted_norm == 'inf':
        nrm = pts.abs().amax(dim=-1, keepdim=True)
    elif contracted_norm == 'l2':
        nrm = pts.norm(dim=-1, keepdim=True)
    else:
        raise NotImplementedError(contracted_norm)
    B = 1 + bg_len
    A = B * 1.0 - 1.0
    inner = nrm <= 1.0
    pts = torch.where(inner, pts, pts / nrm * (B - A / nrm))
    return pts, inner.squeeze(-1)


def viewdir_embedding(viewdirs, viewbase_pe):
    """[R,3] -> [R, 3+6*pe]: [v, sin(v_x f0..), ..., cos(...)] (FourierGrid_model.py:231,632-633)."""
    freq = torch.tensor([float(2 ** i) for i in range(viewbase_pe)])
    e = (viewdirs.unsqueeze(-1) * freq).flatten(-2)
    return torch.cat([viewdirs, e.sin(), e.cos()], -1)


def rgbnet_apply(weights, biases, x):
    """Linear -> ReLU -> ... -> Linear (no final activation) (FourierGrid_model.py:233-241)."""
    h = x
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = F.linear(h, w, b)
        if i + 1 < len(weights):
            h = torch.relu(h)
    return h


def act_shift_from_alpha_init(alpha_init):
    """FourierGrid_model.py:173 (np.log in double, stored as a FloatTensor)."""
    return torch.tensor([math.log(1 / (1 - alpha_init) - 1)], dtype=torch.float32)


@torch.no_grad()
def fouriergrid_render(state, rays_o, rays_d, viewdirs, stepsize, render_depth=True, return_margin=False):
    """Render-mode FourierGridModel.forward (FourierGrid_model.py:554-672).

    `state` keys: density_grid [P,1,G,G,G], k0_grid [P,C,G,G,G], rgbnet_weights/biases (lists, may be
    empty => rgb = sigmoid(k0), C==3), scene_center[3], scene_radius[3], xyz_min[3], xyz_max[3] (the
    contracted bounds -1-bg..1+bg), bg_len, fourier_freq_num, viewbase_pe, act_shift (float or tensor),
    voxel_size_ratio (float), fast_color_thres, contracted_norm, world_len (density grid side).
    Returns the reference's dict.  With return_margin=True also 'margin' [R]: the smallest relative
    distance of any thresholded quantity (alpha>thres, w>thres, T<1e-3) of that ray to its threshold --
    used by tests to tell genuine mismatches from 1-ulp threshold flips.
    """
    R = rays_o.shape[0]
    F_num = int(state['fourier_freq_num'])
    thres = float(state['fast_color_thres'])
    t = sample_t(int(state['world_len']), stepsize, float(state['bg_len']))
    S = t.numel()
    pts, inner = contracted_sample_ray(rays_o, rays_d, state['scene_center'], state['scene_radius'], t,
                                       float(state['bg_len']), state.get('contracted_norm', 'inf'))
    # reference: python float * 0-d fp32 tensor -> fp32 product, then float() (FourierGrid_model.py:572)
    interval = float(torch.tensor(float(state['voxel_size_ratio']), dtype=torch.float32) * stepsize)
    ray_id = torch.arange(R).view(-1, 1).expand(R, S).flatten()
    step_id = torch.arange(S).view(1, -1).expand(R, S).flatten()
    tt = t[None].repeat(R, 1)

    density = fourier_grid_query(state['density_grid'], pts, state['xyz_min'], state['xyz_max'], F_num)
    shift = float(state['act_shift'])
    alpha = ref_ops.raw2alpha(density.flatten().contiguous(), shift, interval)[1].reshape(density.shape)

    margin = torch.full((R,), float('inf'))
    if return_margin and thres > 0:
        m = ((alpha - thres).abs() / thres).amin(dim=1)
        margin = torch.minimum(margin, m)

    if thres > 0:
        m1 = alpha > thres
        pts, tt, density, alpha = pts[m1], tt[m1], density[m1], alpha[m1]
        ray_id, step_id = ray_id[m1.flatten()], step_id[m1.flatten()]
    else:
        pts, tt, density, alpha = pts.reshape(-1, 3), tt.flatten(), density.flatten(), alpha.flatten()

    weights, T, alphainv_last, i_start, i_end = ref_ops.alpha2weight(alpha.contiguous(), ray_id.contiguous(), R)
    if return_margin and alpha.numel() > 0:
        # transmittance after each sample vs the 1e-3 early-stop threshold (only where the scan ran)
        T_after = (T.double() * (1.0 - alpha.double())).float()
        pos = torch.arange(alpha.numel())
        ran = (pos >= i_start[ray_id]) & (pos < i_end[ray_id])
        mt = torch.where(ran, (T_after - 1e-3).abs() / 1e-3, torch.full_like(T_after, float('inf')))
        margin = margin.index_reduce_(0, ray_id, mt, 'amin', include_self=True)
        if thres > 0:
            mw = torch.where(ran, (weights - thres).abs() / thres, torch.full_like(weights, float('inf')))
            margin = margin.index_reduce_(0, ray_id, mw, 'amin', include_self=True)

    if thres > 0:
        m2 = weights > thres
        pts, tt, density, alpha, weights = pts[m2], tt[m2], density[m2], alpha[m2], weights[m2]
        ray_id, step_id = ray_id[m2], step_id[m2]

    k0 = fourier_grid_query(state['k0_grid'], pts, state['xyz_min'], state['xyz_max'],
                            F_num if state['k0_grid'].shape[0] > 1 else 0)
    if k0.dim() == 1:
        k0 = k0.unsqueeze(-1)
    if len(state['rgbnet_weights']) == 0:
        rgb = torch.sigmoid(k0)
    else:
        emb = viewdir_embedding(viewdirs, int(state['viewbase_pe']))[ray_id]
        rgb = torch.sigmoid(rgbnet_apply(state['rgbnet_weights'], state['rgbnet_biases'], torch.cat([k0, emb], -1)))

    rgb_marched = torch.zeros(R, 3).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
    s = 1 - 1 / (1 + tt)
    out = {
        'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
        'raw_density': density, 'raw_alpha': alpha, 'raw_rgb': rgb, 'ray_id': ray_id, 'step_id': step_id,
        'n_max': S, 't': tt, 's': s,
    }
    if render_depth:
        out['depth'] = torch.zeros(R).index_add_(0, ray_id, weights * s)
    if return_margin:
        out['margin'] = margin
    return out


def state_from_reference_model(model):
    """Flatten a reference FourierGridModel (build container only) into the plain `state` dict."""
    ws, bs = [], []
    if model.rgbnet is not None:
        for mod in model.rgbnet.modules():
            if isinstance(mod, torch.nn.Linear):
                ws.append(mod.weight.detach().clone())
                bs.append(mod.bias.detach().clone())
    return {
        'density_grid': model.density.grid.detach().clone(),
        'k0_grid': model.k0.grid.detach().clone(),
        'rgbnet_weights': ws, 'rgbnet_biases': bs,
        'scene_center': model.scene_center.clone(), 'scene_radius': model.scene_radius.clone(),
        'xyz_min': model.xyz_min.clone(), 'xyz_max': model.xyz_max.clone(),
        'bg_len': float(model.bg_len), 'fourier_freq_num': int(model.fourier_freq_num),
        'viewbase_pe': int(model.rgbnet_kwargs['viewbase_pe']),
        'act_shift': float(model.act_shift), 'voxel_size_ratio': float(model.voxel_size_ratio_density),
        'fast_color_thres': float(model.fast_color_thres), 'contracted_norm': model.contracted_norm,
        'world_len': int(model.world_len_density),
    }


# ---------------------------------------------------------------------------------------------
# Training-mode forward (autograd), device-agnostic: used by tests to run ONE reference-shaped
# train step (FourierGrid/run_train.py:185-296) on CPU with the oracle ops and on the GPU with the
# product's drop-in ops, and compare the updated parameters.
# ---------------------------------------------------------------------------------------------
def make_autograd_ops(backend):
    """Raw2Alpha / Alphas2Weights autograd Functions (dvgo.py:430-488) over an extension-module backend
    (`backend.raw2alpha`, `.raw2alpha_backward`, `.alpha2weight`, `.alpha2weight_backward`)."""

    class Raw2Alpha(torch.autograd.Function):
        @staticmethod
        def forward(ctx, density, shift, interval):
            exp, alpha = backend.raw2alpha(density, shift, interval)
            if density.requires_grad:
                ctx.save_for_backward(exp)
                ctx.interval = interval
            return alpha

        @staticmethod
        @torch.autograd.function.once_differentiable
        def backward(ctx, grad_back):
            return backend.raw2alpha_backward(ctx.saved_tensors[0], grad_back.contiguous(), ctx.interval), None, None

    class Alphas2Weights(torch.autograd.Function):
        @staticmethod
        def forward(ctx, alpha, ray_id, N):
            weights, T, alphainv_last, i_start, i_end = backend.alpha2weight(alpha, ray_id, N)
            if alpha.requires_grad:
                ctx.save_for_backward(alpha, weights, T, alphainv_last, i_start, i_end)
                ctx.n_rays = N
            return weights, alphainv_last

        @staticmethod
        @torch.autograd.function.once_differentiable
        def backward(ctx, grad_weights, grad_last):
            alpha, weights, T, alphainv_last, i_start, i_end = ctx.saved_tensors
            grad = backend.alpha2weight_backward(alpha, weights, T, alphainv_last, i_start, i_end, ctx.n_rays,
                                                 grad_weights.contiguous(), grad_last.contiguous())
            return grad, None, None

    return Raw2Alpha, Alphas2Weights


def fouriergrid_train_forward(params, cfg, rays_o, rays_d, viewdirs, stepsize, Raw2Alpha, Alphas2Weights,
                              grid_query=None):
    """FourierGridModel.forward in training mode (FourierGrid_model.py:554-672) on params' device.
    params: dict of leaf tensors density_grid, k0_grid, w0,b0,w1,b1,w2,b2; cfg: the non-learned `state` fields.
    grid_query: differentiable lookup with fourier_grid_query's signature (default: the torch restatement)."""
    fourier_grid_query_ = grid_query or fourier_grid_query
    dev = rays_o.device
    R = rays_o.shape[0]
    F_num, thres = int(cfg['fourier_freq_num']), float(cfg['fast_color_thres'])
    t = sample_t(int(cfg['world_len']), stepsize, float(cfg['bg_len'])).to(dev)
    S = t.numel()
    pts, _ = contracted_sample_ray(rays_o, rays_d, cfg['scene_center'].to(dev), cfg['scene_radius'].to(dev), t,
                                   float(cfg['bg_len']), cfg.get('contracted_norm', 'inf'))
    interval = float(torch.tensor(float(cfg['voxel_size_ratio']), dtype=torch.float32) * stepsize)
    ray_id = torch.arange(R, device=dev).view(-1, 1).expand(R, S).flatten()
    lo, hi = cfg['xyz_min'].to(dev), cfg['xyz_max'].to(dev)
    density = fourier_grid_query_(params['density_grid'], pts, lo, hi, F_num)
    alpha = Raw2Alpha.apply(density.flatten(), float(cfg['act_shift']), interval).reshape(density.shape)
    m1 = alpha > thres
    pts, alpha, tt = pts[m1], alpha[m1], t[None].repeat(R, 1)[m1]
    ray_id = ray_id[m1.flatten()]
    weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, R)
    m2 = weights > thres
    pts, weights, ray_id, tt = pts[m2], weights[m2], ray_id[m2], tt[m2]
    k0 = fourier_grid_query_(params['k0_grid'], pts, lo, hi, F_num)
    emb = viewdir_embedding(viewdirs.cpu(), int(cfg['viewbase_pe'])).to(dev)[ray_id]
    rgb = torch.sigmoid(rgbnet_apply([params['w0'], params['w1'], params['w2']],
                                     [params['b0'], params['b1'], params['b2']], torch.cat([k0, emb], -1)))
    rgb_marched = torch.zeros(R, 3, device=dev).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
    return {'rgb_marched': rgb_marched, 'alphainv_last': alphainv_last, 'weights': weights, 'ray_id': ray_id,
            'n_kept': int(weights.numel())}


# ---------------------------------------------------------------------------------------------
# Bounded DVGO forward (config 1): dvgo.DirectVoxGO.forward, /root/reference/FourierGrid/dvgo.py:306-425.
# Device-agnostic composition: `ops` is an extension-module backend (oracle on CPU, the product's drop-in
# module on the GPU), `query` a dense-grid query with fourier_grid_query's signature.
# ---------------------------------------------------------------------------------------------
def dvgo_state_from_params(xyz_min, xyz_max, num_voxels, num_voxels_base, alpha_init, density_grid, k0_grid,
                           rgbnet_weights, rgbnet_biases, mask, fast_color_thres, rgbnet_direct, viewbase_pe=4):
    """Derived quantities exactly as DirectVoxGO.__init__/_set_grid_resolution compute them (dvgo.py:40-56,154-163)
    and as grid.MaskGrid does (grid.py:221-228)."""
    lo, hi = torch.Tensor(xyz_min), torch.Tensor(xyz_max)
    voxel_size_base = ((hi - lo).prod() / num_voxels_base).pow(1 / 3)
    voxel_size = ((hi - lo).prod() / num_voxels).pow(1 / 3)
    world_size = ((hi - lo) / voxel_size).long()
    scale = (torch.Tensor(list(mask.shape)) - 1) / (hi - lo)
    return {
        'xyz_min': lo, 'xyz_max': hi, 'voxel_size': voxel_size, 'voxel_size_ratio': voxel_size / voxel_size_base,
        'world_size': world_size, 'act_shift': act_shift_from_alpha_init(alpha_init),
        'density_grid': density_grid, 'k0_grid': k0_grid, 'rgbnet_weights': rgbnet_weights,
        'rgbnet_biases': rgbnet_biases, 'mask': mask.bool(), 'xyz2ijk_scale': scale, 'xyz2ijk_shift': -lo * scale,
        'fast_color_thres': fast_color_thres, 'rgbnet_direct': rgbnet_direct, 'viewbase_pe': viewbase_pe,
    }


@torch.no_grad()
def dvgo_render(state, rays_o, rays_d, viewdirs, near, stepsize, bg, ops, query, render_depth=True):
    dev = rays_o.device
    st = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in state.items()}
    N = rays_o.shape[0]
    far = 1e9  # dvgo.py:318
    stepdist = stepsize * st['voxel_size']
    ray_pts, mask_outbbox, ray_id, step_id = ops.sample_pts_on_rays(
        rays_o.contiguous(), rays_d.contiguous(), st['xyz_min'], st['xyz_max'], near, far, stepdist)[:4]
    inb = ~mask_outbbox
    ray_pts, ray_id, step_id = ray_pts[inb], ray_id[inb], step_id[inb]
    interval = stepsize * st['voxel_size_ratio']
    m = ops.maskcache_lookup(st['mask'], ray_pts.contiguous(), st['xyz2ijk_scale'], st['xyz2ijk_shift'])
    ray_pts, ray_id, step_id = ray_pts[m], ray_id[m], step_id[m]
    density = query(st['density_grid'], ray_pts, st['xyz_min'], st['xyz_max'], 0)
    alpha = ops.raw2alpha(density.flatten().contiguous(), st['act_shift'], interval)[1]
    thres = float(st['fast_color_thres'])
    if thres > 0:
        k = alpha > thres
        ray_pts, ray_id, step_id, alpha = ray_pts[k], ray_id[k], step_id[k], alpha[k]
    weights, _, alphainv_last = ops.alpha2weight(alpha.contiguous(), ray_id.contiguous(), N)[:3]
    if thres > 0:
        k = weights > thres
        weights, alpha, ray_pts, ray_id, step_id = weights[k], alpha[k], ray_pts[k], ray_id[k], step_id[k]
    k0 = query(st['k0_grid'], ray_pts, st['xyz_min'], st['xyz_max'], 0)
    if k0.dim() == 1:
        k0 = k0.unsqueeze(-1)
    if len(st['rgbnet_weights']) == 0:
        rgb = torch.sigmoid(k0)
    else:
        emb = viewdir_embedding(viewdirs.cpu(), int(st['viewbase_pe'])).to(dev)[ray_id]
        if st['rgbnet_direct']:
            rgb = torch.sigmoid(rgbnet_apply(st['rgbnet_weights'], st['rgbnet_biases'], torch.cat([k0, emb], -1)))
        else:
            logit = rgbnet_apply(st['rgbnet_weights'], st['rgbnet_biases'], torch.cat([k0[:, 3:], emb], -1))
            rgb = torch.sigmoid(logit + k0[:, :3])
    rgb_marched = torch.zeros(N, 3, device=dev).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
    rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * bg
    out = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha,
           'raw_rgb': rgb, 'ray_id': ray_id}
    if render_depth:
        out['depth'] = torch.zeros(N, device=dev).index_add_(0, ray_id, weights * step_id)
    return out
