"""DirectContractedVoxGORenderer: inference forward of the reference's contracted-unbounded DVGOv2 model
dcvgo.DirectContractedVoxGO (/root/reference/FourierGrid/dcvgo.py:228-384, the "dcvgo contracted bg grid" of
BASELINE.json configs[1]) composed from the drop-in kernels, like DirectVoxGORenderer is for the bounded model:

  mid-point sample table (t_boundary = 2) -> contraction -> cumdist_thres (drops oversampled contracted points)
  -> maskcache_lookup -> dense grid query -> raw2alpha -> alpha2weight -> k0 query -> rgbnet -> per-ray sums
  (+ alphainv_last * bg, wsum_mid over the un-contracted samples, depth = sum w * s)

Same call signature and return keys as the reference forward.  Boolean compactions and the tiny rgbnet use torch on
the device, exactly like the reference; every kernel the reference has natively is the HIP one.  `ops` / `query`
exist for tests: they let the same composition run against another implementation of the four extension modules
(the CPU oracle); the default is the HIP library, which needs a GPU -- there is no CPU fallback.
"""
import torch
import torch.nn.functional as F


def dcvgo_state_from_params(xyz_min, xyz_max, num_voxels, num_voxels_base, alpha_init, density_grid, k0_grid,
                            rgbnet_weights, rgbnet_biases, mask, fast_color_thres, bg_len=0.2, contracted_norm='inf',
                            viewbase_pe=4):
    """The derived quantities of DirectContractedVoxGO.__init__ / _set_grid_resolution (dcvgo.py:42-60,128-136) and of
    its MaskGrid (grid.py:221-228), from the constructor arguments and the learned tensors."""
    import math
    lo_s, hi_s = torch.Tensor(xyz_min), torch.Tensor(xyz_max)            # fg / bg separating cube (scene space)
    lo = torch.Tensor([-1, -1, -1]) - bg_len                              # contracted bounds
    hi = torch.Tensor([1, 1, 1]) + bg_len
    voxel_size_base = ((hi - lo).prod() / num_voxels_base).pow(1 / 3)
    voxel_size = ((hi - lo).prod() / num_voxels).pow(1 / 3)
    world_size = ((hi - lo) / voxel_size).long()
    scale = (torch.Tensor(list(mask.shape)) - 1) / (hi - lo)
    return {
        'scene_center': (lo_s + hi_s) * 0.5, 'scene_radius': (hi_s - lo_s) * 0.5, 'xyz_min': lo, 'xyz_max': hi,
        'bg_len': bg_len, 'contracted_norm': contracted_norm, 'world_size': world_size,
        'world_len': int(world_size[0]), 'voxel_size_ratio': voxel_size / voxel_size_base,
        'act_shift': torch.FloatTensor([math.log(1 / (1 - alpha_init) - 1)]),
        'density_grid': density_grid, 'k0_grid': k0_grid, 'rgbnet_weights': rgbnet_weights,
        'rgbnet_biases': rgbnet_biases, 'mask': mask.bool(), 'xyz2ijk_scale': scale, 'xyz2ijk_shift': -lo * scale,
        'fast_color_thres': fast_color_thres, 'viewbase_pe': viewbase_pe,
    }


def dcvgo_state_from_reference_checkpoint(ckpt):
    """`state` from a checkpoint the reference's trainer wrote for a DirectContractedVoxGO model ({'model_kwargs',
    'model_state_dict'}, dcvgo.py:139-155 get_kwargs): dense grids only; bg_len is not among the saved kwargs (the constructor
    default 0.2 applies on reload, dcvgo.py:29)."""
    kw, sd = ckpt['model_kwargs'], ckpt['model_state_dict']
    if kw.get('density_type', 'DenseGrid') != 'DenseGrid' or kw.get('k0_type', 'DenseGrid') != 'DenseGrid':
        raise NotImplementedError("only DenseGrid checkpoints (TensoRFGrid is outside the hot path, SURVEY.md section 8)")
    if kw.get('rgbnet_full_implicit', False):
        raise NotImplementedError("rgbnet_full_implicit models have no feature grid")
    lin = sorted({k[:-len('.weight')] for k in sd if k.startswith('rgbnet.') and k.endswith('.weight')},
                 key=lambda n: [int(x) for x in n.split('.')[1:]])
    st = dcvgo_state_from_params(
        [float(x) for x in kw['xyz_min']], [float(x) for x in kw['xyz_max']], kw['num_voxels'], kw['num_voxels_base'], kw['alpha_init'],
        sd['density.grid'], sd['k0.grid'], [sd[n + '.weight'] for n in lin], [sd[n + '.bias'] for n in lin], sd['mask_cache.mask'],
        kw.get('fast_color_thres', 0), bg_len=kw.get('bg_len', 0.2), contracted_norm=kw.get('contracted_norm', 'inf'),
        viewbase_pe=kw.get('viewbase_pe', 4))
    for k in ('xyz2ijk_scale', 'xyz2ijk_shift'):          # the stored buffers win over the re-derived ones
        if 'mask_cache.' + k in sd:
            st[k] = sd['mask_cache.' + k]
    return st


class _HipOps:
    """The product's extension modules + grid query, resolved lazily (importing them loads libugrid_hip.so)."""

    def __init__(self):
        from . import render_utils_cuda, ub360_utils_cuda
        from .grid import grid_query
        self.ru, self.ub, self.query = render_utils_cuda, ub360_utils_cuda, grid_query


class DirectContractedVoxGORenderer:
    def __init__(self, state, device, ops=None, query=None):
        dev = torch.device(device)
        if ops is None:
            if dev.type != "cuda":
                raise RuntimeError("DirectContractedVoxGORenderer needs a HIP device (no CPU path)")
            hip = _HipOps()
            self.ru, self.ub, self.query = hip.ru, hip.ub, hip.query
        else:                                   # tests: another implementation of the extension modules
            self.ru, self.ub, self.query = ops.render_utils_cuda, ops.ub360_utils_cuda, query
        self.device = dev
        self.s = {k: (v.to(dev).contiguous() if torch.is_tensor(v) else
                      ([x.to(dev).contiguous() for x in v] if isinstance(v, list) else v)) for k, v in state.items()}
        self.viewfreq = torch.tensor([float(2 ** i) for i in range(int(state["viewbase_pe"]))], device=dev)
        self._tables = {}
        self._fused = None if ops is None else False      # fused render kernels: HIP library only, built on first use

    @classmethod
    def from_reference_checkpoint(cls, ckpt, device):
        """ckpt: the dict the reference saves for a DirectContractedVoxGO model (torch.load('fine_last.tar', weights_only=False))"""
        return cls(dcvgo_state_from_reference_checkpoint(ckpt), device)

    # -- fused inference path ----------------------------------------------------------------------------------
    def fused_supported(self):
        """the fused march (ugrid_render_march_dcvgo) + shade kernels cover: the default HIP ops, fast_color_thres > 0, one
        resolution for both grids, and either no rgbnet (3-channel k0, rgb = sigmoid(k0)) or the 3 x 128 rgbnet on
        [k0 (12), view embedding] that ugrid_shade_supported(0, C, viewbase_pe) lists"""
        if self._fused is False:
            return False
        s = self.s
        if float(s['fast_color_thres']) <= 0 or tuple(s['density_grid'].shape[2:]) != tuple(s['k0_grid'].shape[2:]):
            return False
        C = int(s['k0_grid'].shape[1])
        if len(s['rgbnet_weights']) == 0:
            return C == 3
        from . import _lib
        from .fourier_render import rgbnet_fits_fused
        w = s['rgbnet_weights']
        return (rgbnet_fits_fused(w) and w[0].shape[1] == C + 3 + 6 * int(s['viewbase_pe'])
                and bool(_lib.load().ugrid_shade_supported(0, C, int(s['viewbase_pe']))))

    frames_in_flight = 2      # run_render.render_viewpoints: 1080p frame 6.92 / 5.85 / 6.26 / 5.87 ms at 1 / 2 / 3 / 4 views in flight in the one sweep (the
                              # three-stream run probably had two streams on one hardware queue, profiles/r06/side_stream_queues.txt): nothing beyond two

    def _fused_renderer(self):
        """the fused march + shade renderer over this model's grids (built on first use)"""
        if self._fused is None:
            from .fourier_render import FourierGridRenderer
            s = self.s
            st = {'density_grid': s['density_grid'], 'k0_grid': s['k0_grid'], 'rgbnet_weights': s['rgbnet_weights'],
                  'rgbnet_biases': s['rgbnet_biases'], 'scene_center': s['scene_center'], 'scene_radius': s['scene_radius'],
                  'xyz_min': s['xyz_min'], 'xyz_max': s['xyz_max'], 'bg_len': s['bg_len'], 'fourier_freq_num': 0,
                  'viewbase_pe': s['viewbase_pe'], 'act_shift': float(s['act_shift']), 'voxel_size_ratio': float(s['voxel_size_ratio']),
                  'fast_color_thres': float(s['fast_color_thres']), 'contracted_norm': s['contracted_norm'], 'world_len': s['world_len'],
                  'dcvgo': {'mask': s['mask'], 'xyz2ijk_scale': s['xyz2ijk_scale'], 'xyz2ijk_shift': s['xyz2ijk_shift']}}
            self._fused = FourierGridRenderer(st, self.device)
        return self._fused

    def use_workspace_slot(self, k):
        """Views in flight on two streams take a work list each (run_render.render_viewpoints, FourierGridRenderer.use_workspace_slot);
        False: this model renders through the composed forward, one stream."""
        if not self.fused_supported():
            return False
        self._fused_renderer().use_workspace_slot(k)
        return True

    @torch.no_grad()
    def render_rays(self, rays_o, rays_d, viewdirs, **render_kwargs):
        """Per-ray outputs of forward() -- rgb_marched, depth, alphainv_last, wsum_mid (what the render program consumes,
        run_render.py:46) -- through the FUSED kernels: the whole chain of dcvgo.py:228-384 in two launches, no [N,S,3]
        point tensor, no boolean compactions.  Falls back to forward() for models outside fused_supported().
        render_kwargs as forward(): stepsize, bg, render_depth, plus FourierGridRenderer's ray_order."""
        if not self.fused_supported():
            out = self.forward(rays_o, rays_d, viewdirs, **render_kwargs)
            return {k: out[k] for k in ('rgb_marched', 'depth', 'alphainv_last', 'wsum_mid') if k in out}
        fused = self._fused_renderer()
        kw = dict(render_kwargs)
        if 'bg' in kw and torch.is_tensor(kw['bg']):
            kw['bg'] = kw['bg'].to(self.device)
        out = fused(rays_o.contiguous(), rays_d.contiguous(), viewdirs.contiguous(), **kw)
        return {k: out[k] for k in ('rgb_marched', 'depth', 'alphainv_last', 'wsum_mid') if k in out}

    def _t_table(self, stepsize):
        key = float(stepsize)
        if key not in self._tables:
            from .fourier_render import sample_table
            t, _ = sample_table(self.s['world_len'], key, self.s['bg_len'], t_boundary=2)     # dcvgo.py:243-250
            self._tables[key] = t.to(self.device)
        return self._tables[key]

    def render_view(self, H, W, K, c2w, inverse_y=False, flip_x=False, flip_y=False, **render_kwargs):
        """One whole view through render_rays (fourier_render.render_view_of): {key: [H,W(,3)]} of the per-ray outputs."""
        from .fourier_render import render_view_of
        if not self.fused_supported():      # the composed forward takes no ray_order
            rr = lambda o, d, v, ray_order=None, **kw: self.render_rays(o, d, v, **kw)
        else:
            rr = self.render_rays
        return render_view_of(rr, self.device, H, W, K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y, **render_kwargs)

    @torch.no_grad()
    def forward(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        s = self.s
        assert rays_o.dim() == 2 and rays_o.shape[-1] == 3, 'Only support point queries in [N, 3] format'
        N = rays_o.shape[0]
        stepsize = render_kwargs['stepsize']
        bg_len = s['bg_len']
        t = self._t_table(stepsize)
        S = t.numel()
        # samples in the normalised scene, contracted outside the unit cube / ball (dcvgo.py:240-262)
        o = (rays_o - s['scene_center']) / s['scene_radius']
        d = rays_d / rays_d.norm(dim=-1, keepdim=True)
        pts = o[:, None, :] + d[:, None, :] * t[None, :, None]
        if s['contracted_norm'] == 'inf':
            nrm = pts.abs().amax(dim=-1, keepdim=True)
        elif s['contracted_norm'] == 'l2':
            nrm = pts.norm(dim=-1, keepdim=True)
        else:
            raise NotImplementedError(s['contracted_norm'])
        inner = nrm <= 1
        pts = torch.where(inner, pts, pts / nrm * ((1 + bg_len) - bg_len / nrm))
        inner = inner.squeeze(-1)
        interval = stepsize * s['voxel_size_ratio']
        # keep every un-contracted sample; of the contracted ones only those that moved on by ~one step (:283-289)
        keep = inner.clone()
        dist_thres = (2 + 2 * bg_len) / s['world_len'] * stepsize * 0.95
        dist = (pts[:, 1:] - pts[:, :-1]).norm(dim=-1)
        keep[:, 1:] |= self.ub.cumdist_thres(dist.contiguous(), dist_thres)
        ray_id = torch.arange(N, device=pts.device).view(-1, 1).expand(N, S)[keep]
        step_id = torch.arange(S, device=pts.device).view(1, -1).expand(N, S)[keep]
        tt = t[None].expand(N, S)[keep]
        pts, inner = pts[keep], inner[keep]
        # known free space
        m = self.ru.maskcache_lookup(s['mask'], pts.contiguous(), s['xyz2ijk_scale'], s['xyz2ijk_shift'])
        pts, inner, tt, ray_id, step_id = pts[m], inner[m], tt[m], ray_id[m], step_id[m]
        density = self.query(s['density_grid'], pts, s['xyz_min'], s['xyz_max'], 0)
        alpha = self.ru.raw2alpha(density.flatten().contiguous(), s['act_shift'], interval)[1]
        thres = float(s['fast_color_thres'])
        if thres > 0:
            k = alpha > thres
            pts, inner, tt, ray_id, step_id, density, alpha = pts[k], inner[k], tt[k], ray_id[k], step_id[k], density[k], alpha[k]
        weights, _, alphainv_last = self.ru.alpha2weight(alpha.contiguous(), ray_id.contiguous(), N)[:3]
        if thres > 0:
            k = weights > thres
            pts, inner, tt, ray_id, step_id = pts[k], inner[k], tt[k], ray_id[k], step_id[k]
            density, alpha, weights = density[k], alpha[k], weights[k]
        k0 = self.query(s['k0_grid'], pts, s['xyz_min'], s['xyz_max'], 0)
        if k0.dim() == 1:
            k0 = k0.unsqueeze(-1)
        if len(s['rgbnet_weights']) == 0:
            rgb = torch.sigmoid(k0)
        else:
            e = (viewdirs.unsqueeze(-1) * self.viewfreq).flatten(-2)
            emb = torch.cat([viewdirs, e.sin(), e.cos()], -1)[ray_id]
            h = torch.cat([k0, emb], -1)
            n = len(s['rgbnet_weights'])
            for i in range(n):
                h = F.linear(h, s['rgbnet_weights'][i], s['rgbnet_biases'][i])
                if i + 1 < n:
                    h = torch.relu(h)
            rgb = torch.sigmoid(h)
        dev = pts.device
        rgb_marched = torch.zeros(N, 3, device=dev).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
        rgb_marched += alphainv_last.unsqueeze(-1) * render_kwargs['bg']
        wsum_mid = torch.zeros(N, device=dev).index_add_(0, ray_id[inner], weights[inner])
        sdist = 1 - 1 / (1 + tt)
        out = {'alphainv_last': alphainv_last, 'weights': weights, 'wsum_mid': wsum_mid, 'rgb_marched': rgb_marched,
               'raw_density': density, 'raw_alpha': alpha, 'raw_rgb': rgb, 'ray_id': ray_id, 'step_id': step_id,
               'n_max': S, 't': tt, 's': sdist}
        if render_kwargs.get('render_depth', False):
            out['depth'] = torch.zeros(N, device=dev).index_add_(0, ray_id, weights * sdist)
        return out

    __call__ = forward
