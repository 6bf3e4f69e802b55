"""FourierGridModel for TRAINING on the HIP ops: the counterpart of the reference's
nn.Module of that name (/root/reference/FourierGrid/FourierGrid_model.py:136-672) with the same constructor
arguments, the same parameter / buffer names (`density.grid`, `k0.grid`, `rgbnet.*`, `mask_cache.*`, `act_shift`,
`scene_center`, ... -- state_dicts and `get_kwargs()` checkpoints interchange) and the same methods the training
program calls: `forward` (returns the per-sample dict run_train.py consumes), `scale_volume_grid`,
`update_occupancy_cache`, `density_total_variation_add_grad`, `k0_total_variation_add_grad`, `activate_density`.

Every grid lookup (forward and backward), raw2alpha, alpha2weight, the TV gradient and the mask-cache lookup run on
libugrid_hip.so; the compaction masks, the three rgbnet Linear layers and the two resampling ops of the coarse-to-fine
schedule (F.interpolate, F.max_pool3d) are torch, as in the reference.  Inference should use
fourier_render.FourierGridRenderer (fused kernels); this class exists so that training needs nothing but this
package.  `backend` is a test hook (another implementation of the extension modules, e.g. the CPU oracle)."""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import grid as _grid
from . import ops as _ops


def _hip_backend():
    from . import ops, render_utils_cuda, total_variation_cuda
    return SimpleNamespace(Raw2Alpha=ops.Raw2Alpha, Alphas2Weights=ops.Alphas2Weights, grid_query=None,
                           total_variation_cuda=None, render_utils_cuda=None)


class FourierGridModel(nn.Module):
    def __init__(self, xyz_min, xyz_max, num_voxels_density=0, num_voxels_base_density=0, num_voxels_rgb=0,
                 num_voxels_base_rgb=0, num_voxels_viewdir=-1, alpha_init=None, mask_cache_world_size=None,
                 fast_color_thres=0, bg_len=0.2, contracted_norm='inf', density_type='DenseGrid', k0_type='DenseGrid',
                 density_config={}, k0_config={}, rgbnet_dim=0, rgbnet_depth=3, rgbnet_width=128, fourier_freq_num=5,
                 viewbase_pe=4, img_emb_dim=-1, verbose=False, backend=None, **kwargs):
        super().__init__()
        if num_voxels_viewdir is not None and num_voxels_viewdir > 0:
            raise NotImplementedError("view-direction colour grid (num_voxels_viewdir > 0) is not on the hot path")
        if img_emb_dim > 0 and kwargs.get('sample_num', -1) > 0:
            raise NotImplementedError("per-image appearance embeddings are not on the hot path")
        self._be = backend if backend is not None else _hip_backend()
        # fused stage 1 of the training forward (grid.TrainMarch): on by default with the HIP ops; the composed torch-op
        # chain below remains for injected back-ends, fast_color_thres == 0 and as the A/B reference of the tests
        self.fused_forward = backend is None
        self.fused_sampling2 = backend is None     # grid.TrainSample: stage 2 of the sampling inside the march as well
        self.channels_last_grids = backend is None and kwargs.get('channels_last_grids', True)
        self.splitk_rgbnet = backend is None       # ops.SplitKLinear weight gradients (training on the GPU only)
        self.fused_rgbnet = backend is None        # ops.FusedRgbnet: the default 3 x 128 rgbnet fwd / bwd on the MFMA kernels
        self.fused_loss = backend is None          # train_step.train_iteration: compositing + loss as ops.RenderLoss
        self.native_step = backend is None         # native_step.VoxGOStep: training forward + loss as ONE autograd node issued from C
        self.native_sync_free = False              # True / {'capacity': rows}: that node without its host read (capacity-sized per-sample
                                                   # arrays, counts on the device; native_step.VoxGOStep pack['sync_free'])
        lo_s, hi_s = torch.Tensor(xyz_min), torch.Tensor(xyz_max)
        self.register_buffer('scene_center', (lo_s + hi_s) * 0.5)
        self.register_buffer('scene_radius', (hi_s - lo_s) * 0.5)
        self.register_buffer('xyz_min', torch.Tensor([-1, -1, -1]) - bg_len)      # contracted bounds
        self.register_buffer('xyz_max', torch.Tensor([1, 1, 1]) + bg_len)
        self._fast_color_thres = fast_color_thres if isinstance(fast_color_thres, dict) else None
        self.fast_color_thres = fast_color_thres[0] if isinstance(fast_color_thres, dict) else fast_color_thres
        self.bg_len, self.contracted_norm, self.verbose = bg_len, contracted_norm, verbose
        self.fourier_freq_num = fourier_freq_num
        self.num_voxels_viewdir = num_voxels_viewdir
        self.num_voxels_base_density, self.num_voxels_base_rgb = num_voxels_base_density, num_voxels_base_rgb
        vol = (self.xyz_max - self.xyz_min).prod()
        self.voxel_size_base_density = (vol / num_voxels_base_density).pow(1 / 3)
        self.voxel_size_base_rgb = (vol / num_voxels_base_rgb).pow(1 / 3)
        self._set_grid_resolution(num_voxels_density, num_voxels_rgb)
        self.alpha_init = alpha_init
        self.register_buffer('act_shift', torch.FloatTensor([np.log(1 / (1 - alpha_init) - 1)]))
        self.density_type, self.k0_type = density_type, k0_type
        self.density_config, self.k0_config = density_config, k0_config
        self.world_size = self.world_size_density
        self.density = self._make_grid(1, self.world_size_density, True)
        self.rgbnet_kwargs = {'rgbnet_dim': rgbnet_dim, 'rgbnet_depth': rgbnet_depth, 'rgbnet_width': rgbnet_width,
                              'viewbase_pe': viewbase_pe}
        self.sample_num = kwargs.get('sample_num', -1)
        self.vd = None
        if rgbnet_dim <= 0:                       # coarse stage: a plain 3-channel colour grid
            self.k0_dim = 3
            self.k0 = self._make_grid(3, self.world_size_rgb, False)
            self.rgbnet = None
        else:                                     # feature grid + shallow MLP on [k0, view-direction embedding]
            self.k0_dim = rgbnet_dim
            self.k0 = self._make_grid(rgbnet_dim, self.world_size_rgb, True)
            self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
            layers = [nn.Linear(3 + 6 * viewbase_pe + rgbnet_dim, rgbnet_width), nn.ReLU(inplace=True)]
            layers += [nn.Sequential(nn.Linear(rgbnet_width, rgbnet_width), nn.ReLU(inplace=True))
                       for _ in range(rgbnet_depth - 2)]
            layers += [nn.Linear(rgbnet_width, 3)]
            self.rgbnet = nn.Sequential(*layers)
            nn.init.constant_(self.rgbnet[-1].bias, 0)
        if mask_cache_world_size is None:
            mask_cache_world_size = self.world_size_density
        self.mask_cache = self._make_mask(torch.ones(list(mask_cache_world_size), dtype=torch.bool))

    # -- construction helpers ------------------------------------------------------------------------------
    def _make_grid(self, channels, world_size, fourier):
        # multi-channel grids are stored channel-last on the HIP ops (grid.FourierGrid: same logical parameter, one
        # 4C-byte run per voxel for the lookup / scatter / TV / Adam kernels); injected back-ends keep the canonical layout
        # (the channel-last TV / fused TV + Adam kernels index with 32 bits: a grid of >= 2^31 elements keeps the canonical
        # layout, whose kernels have a scalar 64-bit path -- ADVICE r2)
        n_levels = (1 + 2 * self.fourier_freq_num) if fourier else 1
        numel = n_levels * channels * int(world_size[0]) * int(world_size[1]) * int(world_size[2])
        cfg = {'channels_last': True} if (self.channels_last_grids and channels > 1 and channels % 4 == 0 and numel < 2 ** 31) else None
        g = _grid.FourierGrid(channels=channels, world_size=world_size, xyz_min=self.xyz_min, xyz_max=self.xyz_max,
                              use_nerf_pos=fourier, fourier_freq_num=self.fourier_freq_num, config=cfg)
        g.query_fn, g.tv_module = self._be.grid_query, self._be.total_variation_cuda
        return g

    def _make_mask(self, mask):
        m = _grid.MaskGrid(path=None, mask=mask, xyz_min=self.xyz_min, xyz_max=self.xyz_max)
        m.lookup_module = self._be.render_utils_cuda
        return m

    def _set_grid_resolution(self, num_voxels_density, num_voxels_rgb):
        self.num_voxels_density, self.num_voxels_rgb = num_voxels_density, num_voxels_rgb
        ext = self.xyz_max - self.xyz_min
        self.voxel_size_density = (ext.prod() / num_voxels_density).pow(1 / 3)
        self.voxel_size_rgb = (ext.prod() / num_voxels_rgb).pow(1 / 3)
        self.world_size_density = (ext / self.voxel_size_density).long()
        self.world_size_rgb = (ext / self.voxel_size_rgb).long()
        self.world_len_density = self.world_size_density[0].item()
        self.world_len_rgb = self.world_size_rgb[0].item()
        self.voxel_size_ratio_density = self.voxel_size_density / self.voxel_size_base_density
        self.voxel_size_ratio_rgb = self.voxel_size_rgb / self.voxel_size_base_rgb

    def get_kwargs(self):
        """What the reference stores as `model_kwargs` in its checkpoints (FourierGrid_model.py:350-373)."""
        return {
            'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(),
            'num_voxels_density': self.num_voxels_density, 'num_voxels_rgb': self.num_voxels_rgb,
            'num_voxels_viewdir': self.num_voxels_viewdir, 'fourier_freq_num': self.fourier_freq_num,
            'num_voxels_base_density': self.num_voxels_base_density, 'num_voxels_base_rgb': self.num_voxels_base_rgb,
            'alpha_init': self.alpha_init, 'voxel_size_ratio_density': self.voxel_size_ratio_density,
            'voxel_size_ratio_rgb': self.voxel_size_ratio_rgb,
            'mask_cache_world_size': list(self.mask_cache.mask.shape), 'fast_color_thres': self.fast_color_thres,
            'contracted_norm': self.contracted_norm, 'density_type': self.density_type, 'k0_type': self.k0_type,
            'density_config': self.density_config, 'k0_config': self.k0_config, 'sample_num': self.sample_num,
            **self.rgbnet_kwargs,
        }

    # -- the pieces run_train.py calls -----------------------------------------------------------------------
    def activate_density(self, density, interval=None):
        interval = interval if interval is not None else self.voxel_size_ratio_density
        return self._be.Raw2Alpha.apply(density.flatten(), self.act_shift, interval).reshape(density.shape)

    def density_total_variation_add_grad(self, weight, dense_mode):
        w = weight * self.world_size_density.max() / 128
        self.density.total_variation_add_grad(w, w, w, dense_mode)

    def k0_total_variation_add_grad(self, weight, dense_mode):
        w = weight * self.world_size_rgb.max() / 128
        self.k0.total_variation_add_grad(w, w, w, dense_mode)

    def _cell_centres(self, shape):
        axes = [torch.linspace(float(self.xyz_min[a]), float(self.xyz_max[a]), int(shape[a])) for a in range(3)]
        return torch.stack(torch.meshgrid(*axes, indexing='ij'), -1).to(self.xyz_min.device)

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels_density, num_voxels_rgb):
        """Coarse-to-fine step (FourierGrid_model.py:421-437): resample both grids trilinearly to the new resolution and
        rebuild the mask cache at that resolution from the old cache and the max-pooled alpha of level 0."""
        self._set_grid_resolution(num_voxels_density, num_voxels_rgb)
        self.density.scale_volume_grid(self.world_size_density)
        self.k0.scale_volume_grid(self.world_size_rgb)
        self.world_size = self.world_size_density
        if np.prod(self.world_size_density.tolist()) <= 256 ** 3:
            xyz = self._cell_centres(self.world_size_density.tolist())
            alpha = F.max_pool3d(self.activate_density(self.density.get_dense_grid()), kernel_size=3, padding=1, stride=1)[0, 0]
            self.mask_cache = self._make_mask(self.mask_cache(xyz) & (alpha > self.fast_color_thres)).to(xyz.device)

    @torch.no_grad()
    def update_occupancy_cache(self):
        """AND the mask cache with (3x3x3 max-pooled alpha at the cache's own vertices > fast_color_thres)
        (FourierGrid_model.py:440-453)."""
        xyz = self._cell_centres(self.mask_cache.mask.shape)
        alpha = self.activate_density(self.density(xyz)[None, None])
        alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1)[0, 0]
        self.mask_cache.mask &= (alpha > self.fast_color_thres)

    def voxel_count_views(self, rays_o_tr, rays_d_tr, imsz, near, far, stepsize, downrate=1, irregular_shape=False):
        """How many training views see each voxel of a plain grid of the density resolution (FourierGrid_model.py:
        392-418): per image the trilinear footprint of its rays' samples is scattered into a zero grid (the lookup's
        backward), a voxel counts as seen when it gathered more than 1."""
        far = 1e9
        dev = self.xyz_min.device
        n_samples = int(np.linalg.norm(self.world_size_density.cpu().numpy().astype(np.float64) + 1) / stepsize) + 1
        rng = torch.arange(n_samples, device=dev)[None].float()
        shape = [1, 1] + self.world_size_density.tolist()
        count = torch.zeros(self.density.get_dense_grid().shape, device=dev)
        query = self._be.grid_query
        if query is None:
            query = _grid.GridQuery.apply
        for o_img, d_img in zip(rays_o_tr.split(imsz), rays_d_tr.split(imsz)):
            ones = torch.zeros(shape, device=dev).requires_grad_(True)
            if irregular_shape:
                o_chunks, d_chunks = o_img.split(10000), d_img.split(10000)
            else:
                o_chunks = o_img[::downrate, ::downrate].to(dev).flatten(0, -2).split(10000)
                d_chunks = d_img[::downrate, ::downrate].to(dev).flatten(0, -2).split(10000)
            for o, d in zip(o_chunks, d_chunks):
                vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
                t_min = torch.minimum((self.xyz_max - o) / vec, (self.xyz_min - o) / vec).amax(-1).clamp(min=near, max=far)
                step = stepsize * self.voxel_size_density * rng
                pts = o[..., None, :] + d[..., None, :] * (t_min[..., None] + step / d.norm(dim=-1, keepdim=True))[..., None]
                query(ones, pts, self.xyz_min, self.xyz_max, 0).sum().backward()
            with torch.no_grad():
                count += (ones.grad > 1)
        return count

    def gather_training_rays(self, data_dict, images, cfg, i_train, cfg_train, poses, HW, Ks, render_kwargs):
        """The reference calls this as a METHOD of the model for its FourierGrid datasets (run_train.py:160-161,
        FourierGrid_model.py:297-333); the implementation is train_rays.gather_training_rays."""
        from .train_rays import gather_training_rays
        return gather_training_rays(self, data_dict, images, cfg, i_train, cfg_train, poses, HW, Ks, render_kwargs)

    # -- forward ---------------------------------------------------------------------------------------------
    def sample_ray(self, ori_rays_o, ori_rays_d, stepsize, **unused):
        """Mid-point samples shared by all rays, contracted outside the unit cube / ball (:509-552)."""
        o = (ori_rays_o - self.scene_center) / self.scene_radius
        d = ori_rays_d / ori_rays_d.norm(dim=-1, keepdim=True)
        n_inner = int(2 / (2 + 2 * self.bg_len) * self.world_len_density / stepsize) + 1
        edge_in = torch.linspace(0, 1.5, n_inner + 1)
        edge_out = 1.5 / torch.linspace(1, 1 / 128, n_inner + 1)
        t = torch.cat([(edge_in[1:] + edge_in[:-1]) * 0.5, (edge_out[1:] + edge_out[:-1]) * 0.5]).to(o.device)
        pts = o[:, None, :] + d[:, None, :] * t[None, :, None]
        if self.contracted_norm == 'inf':
            nrm = pts.abs().amax(dim=-1, keepdim=True)
        elif self.contracted_norm == 'l2':
            nrm = pts.norm(dim=-1, keepdim=True)
        else:
            raise NotImplementedError
        B = 1 + self.bg_len
        A = B * 1.0 - 1.0
        inner = nrm <= 1.0
        pts = torch.where(inner, pts, pts / nrm * (B - A / nrm))
        return pts, inner.squeeze(-1), t

    def _host_consts(self):
        """host copies of scene_center / scene_radius / act_shift for the kernel arguments, refreshed only when the
        buffers change (act_shift moves at the pg_scale steps): no device-to-host read per iteration"""
        ver = (self.act_shift._version, self.scene_center._version, self.scene_radius._version, self.act_shift.data_ptr())
        if getattr(self, '_hc_ver', None) != ver:
            self._hc = (self.scene_center.tolist(), self.scene_radius.tolist(), float(self.act_shift))
            self._hc_ver = ver
        return self._hc

    def sample_table(self, stepsize, device):
        key = (float(stepsize), int(self.world_len_density), str(device))
        cached = getattr(self, '_t_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        t = self._sample_table(stepsize).to(device)
        self._t_cache = (key, t)
        return t

    def _sample_table(self, stepsize):
        """the mid-point sample distances of sample_ray, [S] on `device`"""
        n_inner = int(2 / (2 + 2 * self.bg_len) * self.world_len_density / stepsize) + 1
        edge_in = torch.linspace(0, 1.5, n_inner + 1)
        edge_out = 1.5 / torch.linspace(1, 1 / 128, n_inner + 1)
        return torch.cat([(edge_in[1:] + edge_in[:-1]) * 0.5, (edge_out[1:] + edge_out[:-1]) * 0.5])

    def _native_params(self):
        """The parameters of native_step.VoxGOStep (density grid, k0 grid, the rgbnet's three weights and biases), or None when this
        configuration takes the op-by-op ops: default 3-layer rgbnet, gradients on, every one of those parameters trainable, both
        grids looked up by the HIP kernels (no injected query function)"""
        if not (self.native_step and self.fused_rgbnet and self.rgbnet is not None and torch.is_grad_enabled()):
            return None
        lin = _ops.rgbnet_linears(self.rgbnet)
        if lin is None:
            return None
        params = [self.density.grid, self.k0.grid] + [p for l in lin for p in (l.weight, l.bias)]
        if not all(p.requires_grad for p in params) or self.density.query_fn is not None or self.k0.query_fn is not None \
                or self.density.grid.shape[0] != 1 + 2 * max(self.fourier_freq_num, 0) \
                or self.k0.grid.shape[0] != 1 + 2 * max(self.k0.nerf_pos_num_freq, 0):
            return None
        return params

    def forward(self, rays_o, rays_d, viewdirs, global_step=None, is_train=False, **render_kwargs):
        assert rays_o.dim() == 2 and rays_o.shape[-1] == 3, 'Only support point queries in [N, 3] format'
        if self._fast_color_thres is not None and global_step in self._fast_color_thres:
            self.fast_color_thres = self._fast_color_thres[global_step]
        R = rays_o.shape[0]
        interval = render_kwargs['stepsize'] * self.voxel_size_ratio_density
        fused_sampling = self.fused_forward and self.fast_color_thres > 0 and rays_o.is_cuda
        if fused_sampling and self.fused_sampling2:
            # stages 1 + 2 of the sampling as one op (grid.TrainSample): march with the transmittance recurrence inside, one
            # host read, one compaction -- no Raw2Alpha / Alphas2Weights / nonzero / index_select launches
            dev = rays_o.device
            t = self.sample_table(render_kwargs['stepsize'], dev)
            S = t.numel()
            hc = self._host_consts()
            fl = render_kwargs.get('fused_loss')
            native = self._native_params() if (fl is not None and self.splitk_rgbnet) else None
            if native is not None:
                # the whole forward + loss as ONE autograd node issued from C (native_step.VoxGOStep, mode 'fourier'): the same
                # kernels, sizes and order as the ops below
                from .native_step import VoxGOStep
                cfg = {'act_shift': hc[2], 'interval': float(interval), 'thres': float(self.fast_color_thres), 'scene_center': hc[0],
                       'scene_radius': hc[1], 'bg_len': self.bg_len, 'norm_l2': self.contracted_norm == 'l2',
                       'freq_num': self.fourier_freq_num, 'k0_freq_num': self.k0.nerf_pos_num_freq}
                bg = torch.rand(R, 3, device=dev) if render_kwargs.get('rand_bkgd', False) else None
                pack = {'mode': 'fourier', 'cfg': cfg, 't': t, 'rays_o': rays_o.contiguous(), 'rays_d': rays_d.contiguous(),
                        'viewdirs': viewdirs, 'viewfreq': self.viewfreq, 'xyz_min': self.xyz_min, 'xyz_max': self.xyz_max,
                        'k0_xyz_min': self.k0.xyz_min, 'k0_xyz_max': self.k0.xyz_max, 'mask': None, 'target': fl['target'], 'bg': bg,
                        'coef': fl['coef'], 'sync_free': self.native_sync_free}
                loss, mse = VoxGOStep.apply(*native, pack)
                o = pack['out']
                return {'alphainv_last': o['alphainv_last'], 'weights': o['weights'], 'rgb_marched': o['rgb_marched'],
                        'raw_density': o['raw_density'], 'raw_alpha': o['raw_alpha'], 'raw_logits': o['raw_logits'], 'ray_id': o['ray_id'],
                        'step_id': o['step_id'], 'n_max': S, 't': o['t'], 'loss': loss, 'mse': mse, 'loss_mse': o['loss_mse'],
                        'native': pack}
            pts, density, alpha, weights, alphainv_last, ray_id, step_id, tt = _grid.TrainSample.apply(
                self.density.grid, rays_o.contiguous(), rays_d.contiguous(), t, hc[0], hc[1], self.xyz_min, self.xyz_max,
                self.bg_len, self.contracted_norm == 'l2', hc[2], float(interval), float(self.fast_color_thres),
                self.fourier_freq_num)
        else:
            if fused_sampling:
                # stage 1 in two HIP kernels: no [R,S,3] point tensor, no [R,S] density / alpha, no boolean-index gathers
                dev = rays_o.device
                t = self.sample_table(render_kwargs['stepsize'], dev)
                S = t.numel()
                hc = self._host_consts()
                pts, density, ray_id, step_id, tt = _grid.TrainMarch.apply(
                    self.density.grid, rays_o.contiguous(), rays_d.contiguous(), t, hc[0], hc[1], self.xyz_min, self.xyz_max,
                    self.bg_len, self.contracted_norm == 'l2', hc[2], float(interval), float(self.fast_color_thres),
                    self.fourier_freq_num)
                alpha = self.activate_density(density, interval)
                inner = None
            else:
                pts, inner, t = self.sample_ray(rays_o, rays_d, **render_kwargs)
                S = t.numel()
                dev = pts.device
                ray_id = torch.arange(R, device=dev).view(-1, 1).expand(R, S).flatten()
                step_id = torch.arange(S, device=dev).view(1, -1).expand(R, S).flatten()
                tt = t[None].repeat(R, 1)
                density = self.density(pts)
                alpha = self.activate_density(density, interval)
                if self.fast_color_thres > 0:
                    keep = alpha > self.fast_color_thres
                    pts, inner, tt, density, alpha = pts[keep], inner[keep], tt[keep], density[keep], alpha[keep]
                    ray_id, step_id = ray_id[keep.flatten()], step_id[keep.flatten()]
            weights, alphainv_last = self._be.Alphas2Weights.apply(alpha, ray_id, R)
            if self.fast_color_thres > 0:
                # one nonzero (one host sync) for the seven gathers of FourierGrid_model.py:620-629, instead of one per tensor
                keep = torch.nonzero(weights > self.fast_color_thres).squeeze(1)
                pts, tt, density, alpha, weights = (x.index_select(0, keep) for x in (pts, tt, density, alpha, weights))
                ray_id, step_id = ray_id.index_select(0, keep), step_id.index_select(0, keep)
            else:
                pts, weights = pts.reshape(-1, 3), weights.reshape(-1)
        k0 = self.k0(pts)
        fused_loss = render_kwargs.get('fused_loss')
        if self.rgbnet is None:
            rgb = torch.sigmoid(k0)
        else:
            lin = _ops.rgbnet_linears(self.rgbnet) if (self.fused_rgbnet and k0.is_cuda and torch.is_grad_enabled()) else None
            if lin is not None:
                # the rgbnet and its derivative on the hand-written fp32-MFMA kernels (ops.FusedRgbnet): no library GEMMs; the
                # view embedding rows are formed inside, together with the concatenation (ops.rgbnet_features)
                logits = _ops.FusedRgbnet.apply(k0, _ops.ViewRows(viewdirs, self.viewfreq, ray_id), lin[0].weight, lin[0].bias,
                                                lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias)
            else:
                feat = torch.cat([k0, _ops.rgbnet_features(None, viewdirs, self.viewfreq, ray_id)], -1)
                if self.splitk_rgbnet and feat.is_cuda and torch.is_grad_enabled():
                    logits = _ops.sequential_splitk(self.rgbnet, feat)
                else:
                    logits = self.rgbnet(feat)
            if fused_loss is not None and logits.is_cuda and self.splitk_rgbnet:
                # training tail as ONE op (ops.RenderLoss): sigmoid, compositing, background and the loss terms of
                # run_train.py:254-279.  fused_loss = {'target': [R,3], 'coef': ops.loss_coefficients(...)}
                bg = torch.rand(R, 3, device=dev) if render_kwargs.get('rand_bkgd', False) else None
                loss, mse, rgb_marched = _ops.RenderLoss.apply(logits, weights, alphainv_last, density, ray_id, tt, None,
                                                               fused_loss['target'], bg, fused_loss['coef'])
                return {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_density': density,
                        'raw_alpha': alpha, 'raw_logits': logits, 'ray_id': ray_id, 'step_id': step_id, 'n_max': S, 't': tt,
                        'loss': loss, 'mse': mse}
            rgb = torch.sigmoid(logits)
        rgb_marched = torch.zeros(R, 3, device=dev).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
        if render_kwargs.get('rand_bkgd', False):
            rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * torch.rand_like(rgb_marched)
        s = 1 - 1 / (1 + tt)
        out = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_density': density,
               'raw_alpha': alpha, 'raw_rgb': rgb, 'ray_id': ray_id, 'step_id': step_id, 'n_max': S, 't': tt, 's': s}
        if render_kwargs.get('render_depth', False):
            with torch.no_grad():
                out['depth'] = torch.zeros(R, device=dev).index_add_(0, ray_id, weights * s)
        return out
