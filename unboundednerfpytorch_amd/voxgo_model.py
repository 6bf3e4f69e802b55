"""DirectVoxGO and DirectContractedVoxGO for TRAINING on the HIP ops: the counterparts of the
reference's two dense-grid nn.Modules (/root/reference/FourierGrid/dvgo.py:26-425 -- the bounded model of BASELINE.json
configs[0] -- and dcvgo.py:27-384 -- the contracted-unbounded model of configs[1]) with the same constructor arguments, the
same parameter / buffer names (`density.grid`, `k0.grid`, `rgbnet.*`, `mask_cache.*`, `act_shift`, ...: state_dicts and
`get_kwargs()` checkpoints interchange) and the methods the training program calls (`forward`, `scale_volume_grid`,
`update_occupancy_cache`, `voxel_count_views` / `update_occupancy_cache_lt_nviews`, `maskout_near_cam_vox`,
`density_total_variation_add_grad`, `k0_total_variation_add_grad`, `activate_density`).

The training forward is FUSED like FourierGridModel's: the whole sampling -- sample_pts_on_rays / sample_ray, the box and
mask-cache tests, the density lookup, Raw2Alpha, both thresholds, Alphas2Weights -- is one march + one compaction
(grid.TrainSampleVox), the k0 lookup is the channel-last kernel, the default 3 x 128 rgbnet runs on the fp32-MFMA kernels
(ops.FusedRgbnet); `fused_forward = False` selects the op-by-op chain over the same drop-in ops (the A/B reference of the
tests).  Inference should use dvgo_render.DirectVoxGORenderer / dcvgo_render.DirectContractedVoxGORenderer (fused render
kernels).  There is no CPU path: the ops raise without the HIP library."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import grid as _grid
from . import ops as _ops


def _make_rgbnet(dim0, width, depth):
    net = nn.Sequential(nn.Linear(dim0, width), nn.ReLU(inplace=True),
                        *[nn.Sequential(nn.Linear(width, width), nn.ReLU(inplace=True)) for _ in range(depth - 2)],
                        nn.Linear(width, 3))
    nn.init.constant_(net[-1].bias, 0)
    return net


class _VoxGOBase(nn.Module):
    """What the two models share: one resolution for both grids, the mask cache, the coarse-to-fine step, the TV hooks."""
    fused_forward = True        # grid.TrainSampleVox (needs fast_color_thres > 0, like the reference's own masking branches)
    fused_rgbnet = True         # ops.FusedRgbnet for the default 3-layer rgbnet
    native_step = True          # native_step.VoxGOStep: the fused training forward + loss as ONE autograd node issued from C
    native_sync_free = False    # True / {'capacity': rows}: that node without its host read (capacity-sized per-sample arrays, counts
                                # on the device; native_step.VoxGOStep pack['sync_free'])
                                # (same kernels, same bits; default rgbnet, rgbnet_direct, train_iteration's fused_loss)

    def _init_grids(self, density_type, k0_type, density_config, k0_config, k0_dim, channels_last):
        if density_type != 'DenseGrid' or k0_type != 'DenseGrid':
            raise NotImplementedError("only DenseGrid (TensoRFGrid is outside the hot path, SURVEY.md section 8)")
        self.density_type, self.k0_type = density_type, k0_type
        self.density_config, self.k0_config = density_config, k0_config
        self.density = self._make_grid(1, None)
        self.k0_dim = k0_dim
        self.k0 = self._make_grid(k0_dim, channels_last)

    def _make_grid(self, channels, channels_last):
        ws = self.world_size
        numel = channels * int(ws[0]) * int(ws[1]) * int(ws[2])
        cfg = {'channels_last': True} if (channels_last and channels > 1 and channels % 4 == 0 and numel < 2 ** 31) else None
        return _grid.FourierGrid(channels=channels, world_size=ws, xyz_min=self.xyz_min, xyz_max=self.xyz_max,
                                 use_nerf_pos=False, fourier_freq_num=0, config=cfg)

    def _set_grid_resolution(self, num_voxels):
        self.num_voxels = num_voxels
        ext = self.xyz_max - self.xyz_min
        self.voxel_size = (ext.prod() / num_voxels).pow(1 / 3)
        self.world_size = (ext / self.voxel_size).long()
        self.world_len = self.world_size[0].item()
        self.voxel_size_ratio = self.voxel_size / self.voxel_size_base

    def _vertices(self, shape):
        # a DEVICE linspace between the fp32 buffers, as the reference forms it (dvgo.py:141-146, 187-192 under its CUDA default
        # tensor type): torch's CPU and GPU linspace kernels can differ by an ulp, which moves a mask vertex at a cell boundary
        dev = self.xyz_min.device
        axes = [torch.linspace(self.xyz_min[a], self.xyz_max[a], int(shape[a]), device=dev) for a in range(3)]
        return torch.stack(torch.meshgrid(*axes, indexing='ij'), -1)

    def _new_mask(self, mask):
        return _grid.MaskGrid(path=None, mask=mask, xyz_min=self.xyz_min, xyz_max=self.xyz_max)

    def activate_density(self, density, interval=None):
        interval = interval if interval is not None else self.voxel_size_ratio
        return _ops.Raw2Alpha.apply(density.flatten(), self.act_shift, interval).reshape(density.shape)

    def density_total_variation_add_grad(self, weight, dense_mode):
        w = weight * self.world_size.max() / 128
        self.density.total_variation_add_grad(w, w, w, dense_mode)

    def k0_total_variation_add_grad(self, weight, dense_mode):
        w = weight * self.world_size.max() / 128
        self.k0.total_variation_add_grad(w, w, w, dense_mode)

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels):
        """Coarse-to-fine step (dvgo.py:216-236, dcvgo.py:156-176): both grids resampled trilinearly, the mask cache rebuilt at the
        new resolution from the old cache and the 3x3x3 max-pooled alpha."""
        self._set_grid_resolution(num_voxels)
        self.density.scale_volume_grid(self.world_size)
        self.k0.scale_volume_grid(self.world_size)
        if np.prod(self.world_size.tolist()) <= 256 ** 3:
            xyz = self._vertices(self.world_size.tolist())
            alpha = F.max_pool3d(self.activate_density(self.density.get_dense_grid()), kernel_size=3, padding=1, stride=1)[0, 0]
            self.mask_cache = self._new_mask(self.mask_cache(xyz) & (alpha > self.fast_color_thres)).to(xyz.device)
        self._hc_ver = None

    @torch.no_grad()
    def update_occupancy_cache(self):
        """mask &= (3x3x3 max-pooled alpha at the cache's own vertices > fast_color_thres)  (dvgo.py:238-248, dcvgo.py:178-192)"""
        xyz = self._vertices(self.mask_cache.mask.shape)
        alpha = self.activate_density(self.density(xyz)[None, None])
        alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1)[0, 0]
        self.mask_cache.mask &= (alpha > self.fast_color_thres)

    def __setattr__(self, name, value):
        # a replaced mask cache (scale_volume_grid, a checkpoint loader, user code) may reuse the id and the storage address of the
        # freed one: the host copies of its scale / shift are dropped whenever the attribute is assigned
        if name == 'mask_cache':
            object.__setattr__(self, '_hc_ver', None)
        super().__setattr__(name, value)

    def _apply(self, fn, *a, **k):
        object.__setattr__(self, '_hc_ver', None)      # .to(device) / .float(): new storages under old versions
        return super()._apply(fn, *a, **k)

    def _host_consts(self):
        """host copies of the small buffers the kernel takes by value, refreshed only when they change: no device-to-host read
        per iteration"""
        mc = self.mask_cache
        ver = (self.act_shift._version, self.act_shift.data_ptr(), mc.xyz2ijk_scale.data_ptr(), mc.xyz2ijk_scale._version,
               mc.xyz2ijk_shift._version, id(mc))
        if getattr(self, '_hc_ver', None) != ver:
            self._hc = {'act_shift': float(self.act_shift), 'mask_scale': mc.xyz2ijk_scale.tolist(), 'mask_shift': mc.xyz2ijk_shift.tolist()}
            for k in ('scene_center', 'scene_radius'):
                if hasattr(self, k):
                    self._hc[k] = getattr(self, k).tolist()
            self._hc['box_diag'] = float((self.xyz_max - self.xyz_min).norm())
            self._hc_ver = ver
        return self._hc

    def _step_consts(self, stepsize):
        """(interval, stepdist) of a step size as the float32 products the reference forms (stepsize * voxel_size_ratio,
        stepsize * voxel_size: dvgo.py:320,343), read back once per (stepsize, resolution) instead of once per iteration"""
        key = (float(stepsize), int(self.num_voxels))
        if getattr(self, '_sc_key', None) != key:
            self._sc = (float(stepsize * self.voxel_size_ratio), float(stepsize * self.voxel_size))
            self._sc_key = key
        return self._sc

    # names FourierGridModel uses for its two resolutions (train_step.train_iteration reads them)
    @property
    def world_size_density(self):
        return self.world_size

    @property
    def world_size_rgb(self):
        return self.world_size

    def _can_fuse(self, rays_o):
        return self.fused_forward and self.fast_color_thres > 0 and rays_o.is_cuda

    def _native_params(self):
        """The parameters of native_step.VoxGOStep (density grid, k0 grid, the rgbnet's three weights and biases), or None when this
        configuration takes the op-by-op ops: the native step needs the default 3-layer rgbnet fed by all of k0 (rgbnet_direct),
        gradients on, and every one of those parameters trainable."""
        if not (self.native_step and self.fused_rgbnet and self.rgbnet is not None and torch.is_grad_enabled()
                and getattr(self, 'rgbnet_direct', True)):
            return None
        lin = _ops.rgbnet_linears(self.rgbnet)
        if lin is None:
            return None
        params = [self.density.grid, self.k0.grid] + [p for l in lin for p in (l.weight, l.bias)]
        if not all(p.requires_grad for p in params) or self.density.query_fn is not None or self.k0.query_fn is not None \
                or self.density.grid.shape[0] != 1 or self.k0.grid.shape[0] != 1:
            return None
        return params

    def _const_bg(self, N, value, dev):
        """the constant background colour as [N,3] rows (read-only; kept between steps: one fill launch less per iteration), or None
        for a black background"""
        if float(value) == 0.0:
            return None
        key = (int(N), float(value), str(dev))
        cached = getattr(self, '_bg_rows', None)
        if cached is None or cached[0] != key:
            cached = (key, torch.full((N, 3), float(value), device=dev))
            object.__setattr__(self, '_bg_rows', cached)
        return cached[1]

    def _native_forward(self, params, mode, cfg, t, rays_o, rays_d, viewdirs, fused_loss, bg):
        """The training forward + loss as ONE autograd node issued from C (native_step.VoxGOStep): the reference's return dict with
        loss / mse added, the per-sample arrays detached"""
        from .native_step import VoxGOStep
        pack = {'mode': mode, 'cfg': cfg, 't': t, 'rays_o': rays_o, 'rays_d': rays_d, 'viewdirs': viewdirs, 'viewfreq': self.viewfreq,
                'xyz_min': self.xyz_min, 'xyz_max': self.xyz_max, 'k0_xyz_min': self.k0.xyz_min, 'k0_xyz_max': self.k0.xyz_max,
                'mask': self.mask_cache.mask, 'target': fused_loss['target'], 'bg': bg, 'coef': fused_loss['coef'],
                'sync_free': self.native_sync_free}
        loss, mse = VoxGOStep.apply(*params, pack)
        o = pack['out']
        return {'alphainv_last': o['alphainv_last'], 'weights': o['weights'], 'rgb_marched': o['rgb_marched'], 'raw_alpha': o['raw_alpha'],
                'raw_density': o['raw_density'], 'raw_logits': o['raw_logits'], 'ray_id': o['ray_id'], 'step_id': o['step_id'],
                't': o['t'], 'loss': loss, 'mse': mse, 'loss_mse': o['loss_mse'], 'native': pack}

    def _logits(self, k0_view, viewdirs, ray_id):
        """rgbnet([k0, view embedding]) of the surviving samples: the fp32-MFMA kernels for the default 3-layer net while training"""
        lin = _ops.rgbnet_linears(self.rgbnet) if (self.fused_rgbnet and k0_view.is_cuda and torch.is_grad_enabled()) else None
        if lin is not None:
            rows = _ops.ViewRows(viewdirs, self.viewfreq, ray_id)      # the embedding is formed inside, with the concatenation
            return _ops.FusedRgbnet.apply(k0_view, rows, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias)
        emb = _ops.rgbnet_features(None, viewdirs, self.viewfreq, ray_id)
        return self.rgbnet(torch.cat([k0_view, emb], -1))

    def _fused_tail(self, fused_loss, k0, viewdirs, ray_id, residual, weights, alphainv_last, density, tt, bg, N):
        """Training tail as ONE op (ops.RenderLoss): sigmoid, compositing, background and the loss terms of run_train.py:254-279.
        train_step.train_iteration passes fused_loss = {'target': [N,3], 'coef': ops.loss_coefficients(...)}.  The colour logits are
        the rgbnet's output (+ the diffuse channels of the residual model) or, in the coarse stage, the 3-channel k0 itself."""
        if self.rgbnet is None:
            logits = k0
        elif residual:
            logits = self._logits(k0[:, 3:].contiguous(), viewdirs, ray_id) + k0[:, :3]
        else:
            logits = self._logits(k0, viewdirs, ray_id)
        dens = density if density is not None else torch.zeros_like(weights)
        loss, mse, rgb_marched = _ops.RenderLoss.apply(logits.contiguous(), weights, alphainv_last, dens, ray_id, tt, None,
                                                       fused_loss['target'], bg, fused_loss['coef'])
        return logits, loss, mse, rgb_marched

    def _colour(self, k0, viewdirs, ray_id, residual):
        """rgb of the surviving samples (dvgo.py:377-398, dcvgo.py:332-344)"""
        if self.rgbnet is None:
            return torch.sigmoid(k0)
        if residual:
            return torch.sigmoid(self._logits(k0[:, 3:].contiguous(), viewdirs, ray_id) + k0[:, :3])
        return torch.sigmoid(self._logits(k0, viewdirs, ray_id))


class DirectVoxGO(_VoxGOBase):
    """The bounded model (dvgo.py:26-425)."""
    fused_loss = True           # train_step.train_iteration: compositing + loss as ops.RenderLoss (no distortion / nearclip term)

    def sample_table(self, stepsize, device):
        """train_iteration sizes the distortion term's interval from the sample table; the bounded model has neither: length 1"""
        return torch.empty(1)

    def __init__(self, xyz_min, xyz_max, num_voxels=0, num_voxels_base=0, alpha_init=None, mask_cache_path=None,
                 mask_cache_thres=1e-3, mask_cache_world_size=None, fast_color_thres=0, density_type='DenseGrid',
                 k0_type='DenseGrid', density_config={}, k0_config={}, rgbnet_dim=0, rgbnet_direct=False,
                 rgbnet_full_implicit=False, rgbnet_depth=3, rgbnet_width=128, viewbase_pe=4, **kwargs):
        super().__init__()
        if rgbnet_full_implicit:
            raise NotImplementedError("rgbnet_full_implicit has no feature grid: not on the hot path")
        self.register_buffer('xyz_min', torch.Tensor(xyz_min))
        self.register_buffer('xyz_max', torch.Tensor(xyz_max))
        self.fast_color_thres = fast_color_thres
        self.num_voxels_base = num_voxels_base
        self.voxel_size_base = ((self.xyz_max - self.xyz_min).prod() / self.num_voxels_base).pow(1 / 3)
        self.alpha_init = alpha_init
        self.register_buffer('act_shift', torch.FloatTensor([np.log(1 / (1 - alpha_init) - 1)]))
        self._set_grid_resolution(num_voxels)
        self.fourier_freq_num, self.use_fourier_grid = 0, False
        self.rgbnet_kwargs = {'rgbnet_dim': rgbnet_dim, 'rgbnet_direct': rgbnet_direct, 'rgbnet_full_implicit': rgbnet_full_implicit,
                              'rgbnet_depth': rgbnet_depth, 'rgbnet_width': rgbnet_width, 'viewbase_pe': viewbase_pe}
        self.rgbnet_full_implicit, self.rgbnet_direct = rgbnet_full_implicit, rgbnet_direct
        # the residual model slices k0[:, :3] / k0[:, 3:] per sample: a channel-last record serves both from one 4C-byte run
        self._init_grids(density_type, k0_type, density_config, k0_config, 3 if rgbnet_dim <= 0 else rgbnet_dim,
                         kwargs.get('channels_last_grids', True))
        if rgbnet_dim <= 0:
            self.rgbnet = None
        else:
            self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
            dim0 = 3 + 6 * viewbase_pe + (self.k0_dim if rgbnet_direct else self.k0_dim - 3)
            self.rgbnet = _make_rgbnet(dim0, rgbnet_width, rgbnet_depth)
        self.mask_cache_path, self.mask_cache_thres = mask_cache_path, mask_cache_thres
        if mask_cache_world_size is None:
            mask_cache_world_size = self.world_size
        if mask_cache_path:
            # the coarse stage's geometry, looked up at this model's mask vertices (dvgo.py:125-136).  The lookup is a HIP kernel:
            # the model is usually built on the host and moved afterwards, so the prior is evaluated on the current HIP device
            if not torch.cuda.is_available():
                raise RuntimeError("mask_cache_path needs a HIP device (the mask-cache lookup has no CPU path)")
            dev = torch.device("cuda", torch.cuda.current_device())
            prior = _grid.MaskGrid(path=mask_cache_path, mask_cache_thres=mask_cache_thres).to(dev)
            mask = prior(self._vertices(mask_cache_world_size).to(dev)).cpu()
        else:
            mask = torch.ones([int(x) for x in mask_cache_world_size], dtype=torch.bool)
        self.mask_cache = self._new_mask(mask)

    def get_kwargs(self):
        """`model_kwargs` of the reference's checkpoints (dvgo.py:147-164)"""
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(), 'num_voxels': self.num_voxels,
                'num_voxels_base': self.num_voxels_base, 'alpha_init': self.alpha_init, 'voxel_size_ratio': self.voxel_size_ratio,
                'mask_cache_path': self.mask_cache_path, 'mask_cache_thres': self.mask_cache_thres,
                'mask_cache_world_size': list(self.mask_cache.mask.shape), 'fast_color_thres': self.fast_color_thres,
                'density_type': self.density_type, 'k0_type': self.k0_type, 'density_config': self.density_config,
                'k0_config': self.k0_config, **self.rgbnet_kwargs}

    @torch.no_grad()
    def maskout_near_cam_vox(self, cam_o, near_clip):
        """density = -100 at the grid vertices closer than near_clip to any camera centre (dvgo.py:166-180)"""
        xyz = self._vertices(self.world_size.tolist())
        nearest = torch.stack([(xyz.unsqueeze(-2) - co.to(xyz.device)).pow(2).sum(-1).sqrt().amin(-1) for co in cam_o.split(100)]).amin(0)
        self.density.get_dense_grid()
        self.density.grid[nearest[None, None] <= near_clip] = -100

    def voxel_count_views(self, rays_o_tr, rays_d_tr, imsz, near, far, stepsize, downrate=1, irregular_shape=False):
        """How many training views see each voxel (dvgo.py:250-276): per image the trilinear footprint of its rays' samples is
        scattered into a zero grid (the lookup's backward); a voxel counts as seen when it gathered more than 1."""
        far = 1e9
        dev = self.xyz_min.device
        n_samples = int(np.linalg.norm(self.world_size.cpu().numpy().astype(np.float64) + 1) / stepsize) + 1
        rng = torch.arange(n_samples, device=dev)[None].float()
        count = torch.zeros(self.density.get_dense_grid().shape, device=dev)
        for o_img, d_img in zip(rays_o_tr.split(imsz), rays_d_tr.split(imsz)):
            ones = torch.zeros([1, 1] + self.world_size.tolist(), device=dev).requires_grad_(True)
            if irregular_shape:
                o_chunks, d_chunks = o_img.split(10000), d_img.split(10000)
            else:
                o_chunks = o_img[::downrate, ::downrate].to(dev).flatten(0, -2).split(10000)
                d_chunks = d_img[::downrate, ::downrate].to(dev).flatten(0, -2).split(10000)
            for o, d in zip(o_chunks, d_chunks):
                o, d = o.to(dev), d.to(dev)
                vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
                rate_a, rate_b = (self.xyz_max - o) / vec, (self.xyz_min - o) / vec
                t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near, max=far)
                step = stepsize * self.voxel_size * rng
                pts = o[..., None, :] + d[..., None, :] * (t_min[..., None] + step / d.norm(dim=-1, keepdim=True))[..., None]
                _grid.GridQuery.apply(ones, pts, self.xyz_min, self.xyz_max, 0).sum().backward()
            with torch.no_grad():
                count += (ones.grad > 1)
        return count

    def hit_coarse_geo(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        """Does a ray pass through known-occupied space? (dvgo.py:291-304)"""
        from . import render_utils_cuda
        far = 1e9
        shape = rays_o.shape[:-1]
        rays_o, rays_d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
        pts, outbbox, ray_id = render_utils_cuda.sample_pts_on_rays(rays_o, rays_d, self.xyz_min, self.xyz_max, near, far,
                                                                    stepsize * self.voxel_size)[:3]
        inb = ~outbbox
        hit = torch.zeros([len(rays_o)], dtype=torch.bool, device=rays_o.device)
        hit[ray_id[inb][self.mask_cache(pts[inb])]] = 1
        return hit.reshape(shape)

    def sample_ray(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        """dvgo.py:306-330: the in-box samples of every ray, near to far: pts [M,3], ray_id [M], step_id [M]"""
        from . import render_utils_cuda
        far = 1e9
        pts, outbbox, ray_id, step_id = render_utils_cuda.sample_pts_on_rays(
            rays_o.contiguous(), rays_d.contiguous(), self.xyz_min, self.xyz_max, near, far, stepsize * self.voxel_size)[:4]
        inb = ~outbbox
        return pts[inb], ray_id[inb], step_id[inb]

    def forward(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        """Volume rendering of N rays (dvgo.py:332-425): the reference's return dict."""
        assert rays_o.dim() == 2 and rays_o.shape[-1] == 3, 'Only support point queries in [N, 3] format'
        N = rays_o.shape[0]
        interval = render_kwargs['stepsize'] * self.voxel_size_ratio
        if self._can_fuse(rays_o):
            hc = self._host_consts()
            interval_f, stepdist = self._step_consts(render_kwargs['stepsize'])
            cfg = {'mode': 'dvgo', 'act_shift': hc['act_shift'], 'interval': interval_f, 'thres': float(self.fast_color_thres),
                   'mask_scale': hc['mask_scale'], 'mask_shift': hc['mask_shift'], 'near': float(render_kwargs['near']), 'far': 1e9,
                   'stepdist': stepdist, 'slots': int(math.ceil(hc['box_diag'] / stepdist)) + 2}
            fl = render_kwargs.get('fused_loss')
            native = self._native_params() if (fl is not None and float(fl['coef'][2]) == 0.0 and float(fl['coef'][4]) == 0.0) else None
            if native is not None:
                bg = self._const_bg(N, render_kwargs['bg'], rays_o.device)
                out = self._native_forward(native, 'dvgo', cfg, None, rays_o.contiguous(), rays_d.contiguous(), viewdirs, fl, bg)
                for k in ('raw_density', 'step_id', 't'):      # (not in DirectVoxGO's return dict, dvgo.py:405-417)
                    out.pop(k)
                return out
            pts, density, alpha, weights, alphainv_last, ray_id, step_id, tt, _ = _grid.TrainSampleVox.apply(
                self.density.grid, rays_o.contiguous(), rays_d.contiguous(), None, self.xyz_min, self.xyz_max, self.mask_cache.mask, cfg)
        else:
            tt = None
            pts, ray_id, step_id = self.sample_ray(rays_o=rays_o, rays_d=rays_d, **render_kwargs)
            if self.mask_cache is not None:
                m = self.mask_cache(pts)
                pts, ray_id, step_id = pts[m], ray_id[m], step_id[m]
            density = self.density(pts)
            alpha = self.activate_density(density, interval)
            if self.fast_color_thres > 0:
                m = alpha > self.fast_color_thres
                pts, ray_id, step_id, density, alpha = pts[m], ray_id[m], step_id[m], density[m], alpha[m]
            weights, alphainv_last = _ops.Alphas2Weights.apply(alpha, ray_id, N)
            if self.fast_color_thres > 0:
                m = weights > self.fast_color_thres
                weights, alpha, pts, ray_id, step_id = weights[m], alpha[m], pts[m], ray_id[m], step_id[m]
        k0 = self.k0(pts)
        if k0.dim() == 1:
            k0 = k0.unsqueeze(-1)
        residual = self.rgbnet is not None and not self.rgbnet_direct
        dev = rays_o.device
        fused_loss = render_kwargs.get('fused_loss')
        if fused_loss is not None and tt is not None and k0.is_cuda and float(fused_loss['coef'][2]) == 0.0 and float(fused_loss['coef'][4]) == 0.0:
            # (the bounded model has no `s` / `t`: configurations with the distortion or nearclip terms take the composed tail and
            # fail there exactly as they do with the reference's DirectVoxGO)
            bg = torch.full((N, 3), float(render_kwargs['bg']), device=dev) if float(render_kwargs['bg']) != 0.0 else None
            logits, loss, mse, rgb_marched = self._fused_tail(fused_loss, k0, viewdirs, ray_id, residual, weights, alphainv_last, density,
                                                              tt, bg, N)
            return {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha,
                    'raw_logits': logits, 'ray_id': ray_id, 'loss': loss, 'mse': mse}
        rgb = self._colour(k0, viewdirs, ray_id, residual=residual)
        rgb_marched = torch.zeros(N, 3, device=dev).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
        rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * render_kwargs['bg']
        out = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha, 'raw_rgb': rgb,
               'ray_id': ray_id}
        if render_kwargs.get('render_depth', False):
            with torch.no_grad():
                out['depth'] = torch.zeros(N, device=dev).index_add_(0, ray_id, weights * step_id)
        return out


class DirectContractedVoxGO(_VoxGOBase):
    """The contracted-unbounded model (dcvgo.py:27-384)."""
    fused_loss = True           # train_step.train_iteration: compositing + loss as ops.RenderLoss (the losses of run_train.py:254-279)

    def __init__(self, xyz_min, xyz_max, num_voxels=0, num_voxels_base=0, alpha_init=None, mask_cache_world_size=None,
                 fast_color_thres=0, bg_len=0.2, contracted_norm='inf', density_type='DenseGrid', k0_type='DenseGrid',
                 density_config={}, k0_config={}, rgbnet_dim=0, rgbnet_depth=3, rgbnet_width=128, viewbase_pe=4, **kwargs):
        super().__init__()
        lo_s, hi_s = torch.Tensor(xyz_min), torch.Tensor(xyz_max)      # the boundary that separates fg and bg
        self.register_buffer('scene_center', (lo_s + hi_s) * 0.5)
        self.register_buffer('scene_radius', (hi_s - lo_s) * 0.5)
        self.register_buffer('xyz_min', torch.Tensor([-1, -1, -1]) - bg_len)
        self.register_buffer('xyz_max', torch.Tensor([1, 1, 1]) + bg_len)
        self._fast_color_thres = fast_color_thres if isinstance(fast_color_thres, dict) else None
        self.fast_color_thres = fast_color_thres[0] if isinstance(fast_color_thres, dict) else fast_color_thres
        self.bg_len, self.contracted_norm = bg_len, contracted_norm
        if contracted_norm not in ('inf', 'l2'):
            raise NotImplementedError(contracted_norm)
        self.num_voxels_base = num_voxels_base
        self.voxel_size_base = ((self.xyz_max - self.xyz_min).prod() / self.num_voxels_base).pow(1 / 3)
        self._set_grid_resolution(num_voxels)
        self.alpha_init = alpha_init
        self.register_buffer('act_shift', torch.FloatTensor([np.log(1 / (1 - alpha_init) - 1)]))
        self.rgbnet_kwargs = {'rgbnet_dim': rgbnet_dim, 'rgbnet_depth': rgbnet_depth, 'rgbnet_width': rgbnet_width,
                              'viewbase_pe': viewbase_pe}
        self._init_grids(density_type, k0_type, density_config, k0_config, 3 if rgbnet_dim <= 0 else rgbnet_dim,
                         kwargs.get('channels_last_grids', True))
        if rgbnet_dim <= 0:
            self.rgbnet = None
        else:
            self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
            self.rgbnet = _make_rgbnet(3 + 6 * viewbase_pe + self.k0_dim, rgbnet_width, rgbnet_depth)
        if mask_cache_world_size is None:
            mask_cache_world_size = self.world_size
        self.mask_cache = self._new_mask(torch.ones([int(x) for x in mask_cache_world_size], dtype=torch.bool))

    def get_kwargs(self):
        """`model_kwargs` of the reference's checkpoints (dcvgo.py:138-154)"""
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(), 'num_voxels': self.num_voxels,
                'num_voxels_base': self.num_voxels_base, 'alpha_init': self.alpha_init, 'voxel_size_ratio': self.voxel_size_ratio,
                'mask_cache_world_size': list(self.mask_cache.mask.shape), 'fast_color_thres': self.fast_color_thres,
                'contracted_norm': self.contracted_norm, 'density_type': self.density_type, 'k0_type': self.k0_type,
                'density_config': self.density_config, 'k0_config': self.k0_config, **self.rgbnet_kwargs}

    def update_occupancy_cache_lt_nviews(self, rays_o_tr, rays_d_tr, imsz, render_kwargs, maskout_lt_nviews):
        """mask &= (voxel seen by at least maskout_lt_nviews training views)  (dcvgo.py:194-214)"""
        dev = self.xyz_min.device
        count = torch.zeros(self.density.get_dense_grid().shape, dtype=torch.long, device=dev)
        for o_img, d_img in zip(rays_o_tr.split(imsz), rays_d_tr.split(imsz)):
            ones = torch.zeros([1, 1] + self.world_size.tolist(), device=dev).requires_grad_(True)
            for o, d in zip(o_img.split(8192), d_img.split(8192)):
                pts = self.sample_ray(ori_rays_o=o.to(dev), ori_rays_d=d.to(dev), **render_kwargs)[0]
                _grid.GridQuery.apply(ones, pts, self.xyz_min, self.xyz_max, 0).sum().backward()
            count += (ones.grad > 1)
        self.mask_cache.mask &= (count >= maskout_lt_nviews)[0, 0]

    def _sample_table(self, stepsize):
        """the mid-point sample distances of sample_ray (dcvgo.py:243-250), [S] on the host"""
        n_inner = int(2 / (2 + 2 * self.bg_len) * self.world_len / stepsize) + 1
        b_inner = torch.linspace(0, 2, n_inner + 1)
        b_outer = 2 / torch.linspace(1, 1 / 128, n_inner + 1)
        return torch.cat([(b_inner[1:] + b_inner[:-1]) * 0.5, (b_outer[1:] + b_outer[:-1]) * 0.5])

    def sample_table(self, stepsize, device):
        key = (float(stepsize), int(self.world_len), str(device))
        cached = getattr(self, '_t_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        t = self._sample_table(stepsize).to(device)
        self._t_cache = (key, t)
        return t

    def sample_ray(self, ori_rays_o, ori_rays_d, stepsize, is_train=False, **render_kwargs):
        """dcvgo.py:228-263: [N,S,3] points (contracted outside the unit cube / ball), inner_mask [N,S], t [S]"""
        o = (ori_rays_o - self.scene_center) / self.scene_radius
        d = ori_rays_d / ori_rays_d.norm(dim=-1, keepdim=True)
        t = self.sample_table(stepsize, o.device)
        pts = o[:, None, :] + d[:, None, :] * t[None, :, None]
        nrm = pts.abs().amax(dim=-1, keepdim=True) if self.contracted_norm == 'inf' else pts.norm(dim=-1, keepdim=True)
        inner = nrm <= 1
        pts = torch.where(inner, pts, pts / nrm * ((1 + self.bg_len) - self.bg_len / nrm))
        return pts, inner.squeeze(-1), t

    def forward(self, rays_o, rays_d, viewdirs, global_step=None, is_train=False, **render_kwargs):
        """Volume rendering of N rays (dcvgo.py:265-384): the reference's return dict."""
        assert rays_o.dim() == 2 and rays_o.shape[-1] == 3, 'Only support point queries in [N, 3] format'
        if self._fast_color_thres is not None and global_step in self._fast_color_thres:
            self.fast_color_thres = self._fast_color_thres[global_step]
        N = rays_o.shape[0]
        stepsize = render_kwargs['stepsize']
        interval = stepsize * self.voxel_size_ratio
        dist_thres = (2 + 2 * self.bg_len) / self.world_len * stepsize * 0.95
        dev = rays_o.device
        if self._can_fuse(rays_o):
            hc = self._host_consts()
            t = self.sample_table(stepsize, dev)
            n_max = t.numel()
            cfg = {'mode': 'dcvgo', 'act_shift': hc['act_shift'], 'interval': self._step_consts(stepsize)[0], 'thres': float(self.fast_color_thres),
                   'mask_scale': hc['mask_scale'], 'mask_shift': hc['mask_shift'], 'scene_center': hc['scene_center'],
                   'scene_radius': hc['scene_radius'], 'bg_len': self.bg_len, 'norm_l2': self.contracted_norm == 'l2',
                   'dist_thres': dist_thres}
            fl = render_kwargs.get('fused_loss')
            native = self._native_params() if fl is not None else None
            if native is not None:
                if render_kwargs.get('rand_bkgd', False) and is_train:
                    bg = torch.rand(N, 3, device=dev)
                else:
                    bg = self._const_bg(N, render_kwargs['bg'], dev)
                out = self._native_forward(native, 'dcvgo', cfg, t, rays_o.contiguous(), rays_d.contiguous(), viewdirs, fl, bg)
                out['n_max'] = n_max
                return out
            pts, density, alpha, weights, alphainv_last, ray_id, step_id, tt, inner = _grid.TrainSampleVox.apply(
                self.density.grid, rays_o.contiguous(), rays_d.contiguous(), t, self.xyz_min, self.xyz_max, self.mask_cache.mask, cfg)
        else:
            from . import ub360_utils_cuda
            pts, inner, t = self.sample_ray(ori_rays_o=rays_o, ori_rays_d=rays_d, is_train=global_step is not None, **render_kwargs)
            n_max = t.numel()
            ray_id = torch.arange(N, device=dev).view(-1, 1).expand(N, n_max)
            step_id = torch.arange(n_max, device=dev).view(1, -1).expand(N, n_max)
            keep = inner.clone()
            dist = (pts[:, 1:] - pts[:, :-1]).norm(dim=-1)
            keep[:, 1:] |= ub360_utils_cuda.cumdist_thres(dist.contiguous(), dist_thres)
            pts, inner, tt, ray_id, step_id = pts[keep], inner[keep], t[None].expand(N, n_max)[keep], ray_id[keep], step_id[keep]
            m = self.mask_cache(pts)
            pts, inner, tt, ray_id, step_id = pts[m], inner[m], tt[m], ray_id[m], step_id[m]
            density = self.density(pts)
            alpha = self.activate_density(density, interval)
            if self.fast_color_thres > 0:
                m = alpha > self.fast_color_thres
                pts, inner, tt, ray_id, step_id, density, alpha = pts[m], inner[m], tt[m], ray_id[m], step_id[m], density[m], alpha[m]
            weights, alphainv_last = _ops.Alphas2Weights.apply(alpha, ray_id, N)
            if self.fast_color_thres > 0:
                m = weights > self.fast_color_thres
                pts, inner, tt, ray_id, step_id = pts[m], inner[m], tt[m], ray_id[m], step_id[m]
                density, alpha, weights = density[m], alpha[m], weights[m]
        k0 = self.k0(pts)
        if k0.dim() == 1:
            k0 = k0.unsqueeze(-1)
        fused_loss = render_kwargs.get('fused_loss')
        if fused_loss is not None and k0.is_cuda:
            if render_kwargs.get('rand_bkgd', False) and is_train:
                bg = torch.rand(N, 3, device=dev)
            else:
                bg = torch.full((N, 3), float(render_kwargs['bg']), device=dev) if float(render_kwargs['bg']) != 0.0 else None
            logits, loss, mse, rgb_marched = self._fused_tail(fused_loss, k0, viewdirs, ray_id, False, weights, alphainv_last, density, tt,
                                                              bg, N)
            return {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_density': density,
                    'raw_alpha': alpha, 'raw_logits': logits, 'ray_id': ray_id, 'step_id': step_id, 'n_max': n_max, 't': tt,
                    'loss': loss, 'mse': mse}
        rgb = self._colour(k0, viewdirs, ray_id, residual=False)
        rgb_marched = torch.zeros(N, 3, device=dev).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
        if render_kwargs.get('rand_bkgd', False) and is_train:
            rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * torch.rand_like(rgb_marched)
        else:
            rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * render_kwargs['bg']
        wsum_mid = torch.zeros(N, device=dev).index_add_(0, ray_id[inner], weights[inner])
        s = 1 - 1 / (1 + tt)
        out = {'alphainv_last': alphainv_last, 'weights': weights, 'wsum_mid': wsum_mid, 'rgb_marched': rgb_marched,
               'raw_density': density, 'raw_alpha': alpha, 'raw_rgb': rgb, 'ray_id': ray_id, 'step_id': step_id, 'n_max': n_max,
               't': tt, 's': s}
        if render_kwargs.get('render_depth', False):
            with torch.no_grad():
                out['depth'] = torch.zeros(N, device=dev).index_add_(0, ray_id, weights * s)
        return out
