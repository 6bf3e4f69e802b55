"""DirectVoxGORenderer: inference forward of the reference's bounded model dvgo.DirectVoxGO
(/root/reference/FourierGrid/dvgo.py:306-425, BASELINE.json configs[0]) composed from the drop-in kernels:
sample_pts_on_rays (ray-AABB clip + variable-length marching) -> maskcache_lookup -> dense grid query ->
raw2alpha -> alpha2weight -> k0 query -> rgbnet -> per-ray sums.  Same call signature and return keys as the
reference forward (`rgb_marched`, `depth`, `alphainv_last`, `weights`, `raw_alpha`, `raw_rgb`, `ray_id`).

The compaction steps (boolean masks) and the tiny rgbnet use torch on the device, exactly like the
reference does; every kernel the reference has natively is the HIP one.

Also the training-ray preparation that leans on the same kernels: `hit_coarse_geo`
(dvgo.py:292-304), `voxel_count_views` (dvgo.py:247-277) and `get_training_rays_in_maskcache_sampling`
(dvgo.py:619-657).  `ops` / `query` / `grad_query` exist for tests: they let the same composition run against another
implementation of the extension modules (the CPU oracle); the default is the HIP library, which needs a GPU.
"""
import numpy as np
import torch
import torch.nn.functional as F


def dvgo_state_from_params(xyz_min, xyz_max, num_voxels, num_voxels_base, alpha_init, density_grid, k0_grid, rgbnet_weights,
                           rgbnet_biases, mask, fast_color_thres, rgbnet_direct, viewbase_pe=4):
    """The renderer's `state` from DirectVoxGO's constructor arguments and learned tensors: voxel sizes and world size as
    __init__ / _set_grid_resolution derive them (dvgo.py:40-56, 154-163), act_shift = log(1/(1-alpha_init) - 1) (:49), the
    world -> mask index map of its MaskGrid (grid.py:221-228).  All fp32 tensor arithmetic, like the reference."""
    import math
    lo, hi = torch.Tensor(xyz_min), torch.Tensor(xyz_max)
    vol = (hi - lo).prod()
    voxel_size = (vol / num_voxels).pow(1 / 3)
    scale = (torch.Tensor(list(mask.shape)) - 1) / (hi - lo)
    return {'xyz_min': lo, 'xyz_max': hi, 'voxel_size': voxel_size, 'voxel_size_ratio': voxel_size / (vol / num_voxels_base).pow(1 / 3),
            'world_size': ((hi - lo) / voxel_size).long(), 'act_shift': torch.FloatTensor([math.log(1 / (1 - alpha_init) - 1)]),
            'density_grid': density_grid, 'k0_grid': k0_grid, 'rgbnet_weights': list(rgbnet_weights),
            'rgbnet_biases': list(rgbnet_biases), 'mask': mask.bool(), 'xyz2ijk_scale': scale, 'xyz2ijk_shift': -lo * scale,
            'fast_color_thres': fast_color_thres, 'rgbnet_direct': bool(rgbnet_direct), 'viewbase_pe': int(viewbase_pe)}


def dvgo_state_from_reference_checkpoint(ckpt):
    """`state` from a checkpoint the reference's trainer wrote for a DirectVoxGO model ({'model_kwargs', 'model_state_dict'},
    run_train.py / utils.load_model): dense grids only (density_type = k0_type = 'DenseGrid', the default)."""
    kw, sd = ckpt['model_kwargs'], ckpt['model_state_dict']
    if kw.get('density_type', 'DenseGrid') != 'DenseGrid' or kw.get('k0_type', 'DenseGrid') != 'DenseGrid':
        raise NotImplementedError("only DenseGrid checkpoints (TensoRFGrid is outside the hot path, SURVEY.md section 8)")
    if kw.get('rgbnet_full_implicit', False):
        raise NotImplementedError("rgbnet_full_implicit models have no feature grid")
    lin = sorted({k[:-len('.weight')] for k in sd if k.startswith('rgbnet.') and k.endswith('.weight')},
                 key=lambda n: [int(x) for x in n.split('.')[1:]])
    st = dvgo_state_from_params(
        [float(x) for x in kw['xyz_min']], [float(x) for x in kw['xyz_max']], kw['num_voxels'], kw['num_voxels_base'], kw['alpha_init'], sd['density.grid'], sd['k0.grid'],
        [sd[n + '.weight'] for n in lin], [sd[n + '.bias'] for n in lin], sd['mask_cache.mask'], kw.get('fast_color_thres', 0),
        kw.get('rgbnet_direct', False), kw.get('viewbase_pe', 4))
    for k in ('xyz2ijk_scale', 'xyz2ijk_shift'):          # the stored buffers win over the re-derived ones
        if 'mask_cache.' + k in sd:
            st[k] = sd['mask_cache.' + k]
    return st


class DirectVoxGORenderer:
    """state: xyz_min/xyz_max [3], density_grid [1,1,X,Y,Z], k0_grid [1,C,X,Y,Z], rgbnet_weights/biases (lists),
    mask [mx,my,mz] bool, xyz2ijk_scale/shift [3], act_shift, voxel_size, voxel_size_ratio (0-d tensors or floats),
    fast_color_thres, rgbnet_direct, viewbase_pe."""

    def __init__(self, state, device, ops=None, query=None, grad_query=None):
        dev = torch.device(device)
        if ops is None:
            if dev.type != "cuda":
                raise RuntimeError("DirectVoxGORenderer needs a HIP device (no CPU path)")
            from . import render_utils_cuda
            from .grid import GridQuery, grid_query
            self.ru, self.query, self.grad_query = render_utils_cuda, grid_query, GridQuery.apply
        else:                                   # tests: another implementation of the extension modules
            self.ru, self.query, self.grad_query = ops.render_utils_cuda, query, grad_query or query
        self.device = dev
        self.s = {k: (v.to(dev).contiguous() if torch.is_tensor(v) else
                      ([x.to(dev).contiguous() for x in v] if isinstance(v, list) else v)) for k, v in state.items()}
        self.viewfreq = torch.tensor([float(2 ** i) for i in range(int(state["viewbase_pe"]))], device=dev)
        self._fused = None if ops is None else False      # fused render kernels: HIP library only, built on first use

    @classmethod
    def from_reference_checkpoint(cls, ckpt, device):
        """ckpt: the dict the reference saves for a DirectVoxGO model (torch.load('fine_last.tar', weights_only=False))"""
        return cls(dvgo_state_from_reference_checkpoint(ckpt), device)

    # -- fused inference path ----------------------------------------------------------------------------------
    def fused_supported(self):
        """the fused march (ugrid_render_march_dvgo) + shade kernels cover: the default HIP ops, fast_color_thres > 0, one
        resolution for both grids, and either no rgbnet (3-channel k0, rgb = sigmoid(k0)) or the DIRECT 3 x 128 rgbnet on
        [k0 (12), view embedding] that ugrid_shade_supported(0, C, viewbase_pe) lists -- rgbnet_direct, or the diffuse + residual
        colour of dvgo.py:385-398 (rgbnet on [k0[3:], embedding], k0[:3] added to its output: the shade kernels' residual epilogue)"""
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
        # rgbnet_direct: the network reads all C channels; else (dvgo.py:385-398) channels 3.. and the first three are added to
        # its output -- the shade kernels' residual epilogue (include/ugrid_hip.h: UGRID_MLP_RESIDUAL), C >= 9
        c_in = C if bool(s['rgbnet_direct']) else C - 3
        return (rgbnet_fits_fused(w) and (bool(s['rgbnet_direct']) or C >= 9)
                and w[0].shape[1] == c_in + 3 + 6 * int(s['viewbase_pe'])
                and bool(_lib.load().ugrid_shade_supported(0, C, int(s['viewbase_pe']))))

    frames_in_flight = 4      # run_render.render_viewpoints: a bounded scene's view (800 x 800 on the lego box: two launches of ~0.9 ms that leave most of
                              # the chip idle) gains from four views in flight -- 1.88 / 1.03 / 0.78 ms per view at 1 / 2 / 4 (profiles/r06/frames_in_flight_sweep.txt)

    def _fused_renderer(self):
        """the fused march + shade renderer over this model's grids (built on first use)"""
        if self._fused is None:
            from .fourier_render import FourierGridRenderer
            s = self.s
            lo, hi = s['xyz_min'], s['xyz_max']
            st = {'density_grid': s['density_grid'], 'k0_grid': s['k0_grid'], 'rgbnet_weights': s['rgbnet_weights'],
                  'rgbnet_biases': s['rgbnet_biases'], 'scene_center': (lo + hi) * 0.5, 'scene_radius': (hi - lo) * 0.5,
                  'xyz_min': lo, 'xyz_max': hi, 'bg_len': 0.0, 'fourier_freq_num': 0, 'viewbase_pe': s['viewbase_pe'],
                  'act_shift': float(s['act_shift']), 'voxel_size_ratio': float(s['voxel_size_ratio']),
                  'fast_color_thres': float(s['fast_color_thres']), 'contracted_norm': 'inf', 'world_len': 0,
                  'rgbnet_residual': (len(s['rgbnet_weights']) > 0 and not bool(s['rgbnet_direct'])),
                  'dvgo': {'mask': s['mask'], 'xyz2ijk_scale': s['xyz2ijk_scale'], 'xyz2ijk_shift': s['xyz2ijk_shift'],
                           'voxel_size': s['voxel_size']}}
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
        """Per-ray outputs of forward() -- rgb_marched, depth, alphainv_last (what the render program consumes,
        run_render.py:46) -- through the FUSED kernels: the whole chain of dvgo.py:306-425 in two launches.  The reference
        (and forward()) size the sample list by a count kernel, a cumsum and a HOST READ of the total before the fill
        (render_utils_kernel.cu:100-260, `.item()` in sample_pts_on_rays); here a lane marches its ray to the ray's own
        step count, so nothing is read back.  Falls back to forward() for models outside fused_supported().
        render_kwargs as forward(): near, stepsize, bg, render_depth, plus FourierGridRenderer's ray_order."""
        if not self.fused_supported():
            out = self.forward(rays_o, rays_d, viewdirs, **render_kwargs)
            return {k: out[k] for k in ('rgb_marched', 'depth', 'alphainv_last') if k in out}
        fused = self._fused_renderer()
        kw = dict(render_kwargs)
        if 'bg' in kw and torch.is_tensor(kw['bg']):
            kw['bg'] = kw['bg'].to(self.device)
        out = fused(rays_o.contiguous(), rays_d.contiguous(), viewdirs.contiguous(), **kw)
        return {k: out[k] for k in ('rgb_marched', 'depth', 'alphainv_last') if k in out}

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
        assert rays_o.dim() == 2 and rays_o.shape[-1] == 3, 'Only suuport point queries in [N, 3] format'
        N = rays_o.shape[0]
        stepsize = render_kwargs['stepsize']
        far = 1e9  # the given far can be too small while rays stop when hitting scene bbox (dvgo.py:318)
        stepdist = stepsize * s['voxel_size']
        ray_pts, mask_outbbox, ray_id, step_id = self.ru.sample_pts_on_rays(
            rays_o.contiguous(), rays_d.contiguous(), s['xyz_min'], s['xyz_max'], render_kwargs['near'], far, stepdist)[:4]
        inb = ~mask_outbbox
        ray_pts, ray_id, step_id = ray_pts[inb], ray_id[inb], step_id[inb]
        interval = stepsize * s['voxel_size_ratio']
        m = self.ru.maskcache_lookup(s['mask'], ray_pts.contiguous(), s['xyz2ijk_scale'], s['xyz2ijk_shift'])
        ray_pts, ray_id, step_id = ray_pts[m], ray_id[m], step_id[m]
        density = self.query(s['density_grid'], ray_pts, s['xyz_min'], s['xyz_max'], 0)
        alpha = self.ru.raw2alpha(density.flatten().contiguous(), s['act_shift'], interval)[1]
        thres = float(s['fast_color_thres'])
        if thres > 0:
            k = alpha > thres
            ray_pts, ray_id, step_id, alpha = ray_pts[k], ray_id[k], step_id[k], alpha[k]
        weights, _, alphainv_last = self.ru.alpha2weight(alpha.contiguous(), ray_id.contiguous(), N)[:3]
        if thres > 0:
            k = weights > thres
            weights, alpha, ray_pts, ray_id, step_id = weights[k], alpha[k], ray_pts[k], ray_id[k], step_id[k]
        k0 = self.query(s['k0_grid'], ray_pts, s['xyz_min'], s['xyz_max'], 0)
        if k0.dim() == 1:
            k0 = k0.unsqueeze(-1)
        if len(s['rgbnet_weights']) == 0:
            rgb = torch.sigmoid(k0)
        else:
            e = (viewdirs.unsqueeze(-1) * self.viewfreq).flatten(-2)
            emb = torch.cat([viewdirs, e.sin(), e.cos()], -1)[ray_id]
            feat = torch.cat([k0 if s['rgbnet_direct'] else k0[:, 3:], emb], -1)
            h = feat
            n = len(s['rgbnet_weights'])
            for i in range(n):
                h = F.linear(h, s['rgbnet_weights'][i], s['rgbnet_biases'][i])
                if i + 1 < n:
                    h = torch.relu(h)
            rgb = torch.sigmoid(h if s['rgbnet_direct'] else h + k0[:, :3])
        rgb_marched = torch.zeros(N, 3, device=rays_o.device).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
        rgb_marched += alphainv_last.unsqueeze(-1) * render_kwargs['bg']
        out = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha,
               'raw_rgb': rgb, 'ray_id': ray_id}
        if render_kwargs.get('render_depth', False):
            out['depth'] = torch.zeros(N, device=rays_o.device).index_add_(0, ray_id, weights * step_id)
        return out

    __call__ = forward

    # -- training-ray preparation ---------------------------------------------------------------------------
    @torch.no_grad()
    def hit_coarse_geo(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        """bool [...]: does the ray pass through a cell the mask cache marks as possibly occupied? (dvgo.py:292-304)"""
        s = self.s
        far = 1e9
        shape = rays_o.shape[:-1]
        o = rays_o.reshape(-1, 3).contiguous()
        d = rays_d.reshape(-1, 3).contiguous()
        ray_pts, mask_outbbox, ray_id = self.ru.sample_pts_on_rays(o, d, s['xyz_min'], s['xyz_max'], near, far,
                                                                  stepsize * s['voxel_size'])[:3]
        inb = ~mask_outbbox
        pts_in, rid_in = ray_pts[inb], ray_id[inb]
        occ = self.ru.maskcache_lookup(s['mask'], pts_in.contiguous(), s['xyz2ijk_scale'], s['xyz2ijk_shift'])
        hit = torch.zeros(o.shape[0], dtype=torch.bool, device=o.device)
        hit[rid_in[occ]] = True
        return hit.reshape(shape)

    def voxel_count_views(self, rays_o_tr, rays_d_tr, imsz, near, far, stepsize, downrate=1, irregular_shape=False):
        """Per-voxel count of the training views that see it (dvgo.py:247-277): for every image the trilinear
        footprint of its rays' samples is scattered into a zero grid -- here by the lookup's scatter backward -- and a
        voxel counts as seen when its accumulated weight exceeds 1."""
        s = self.s
        far = 1e9
        ws = s['world_size']
        dev = s['density_grid'].device
        n_samples = int(np.linalg.norm(ws.cpu().numpy().astype(np.float64) + 1) / stepsize) + 1
        rng = torch.arange(n_samples, device=dev)[None].float()
        count = torch.zeros_like(s['density_grid'])
        for o_img, d_img in zip(rays_o_tr.split(imsz), rays_d_tr.split(imsz)):
            ones = torch.zeros_like(s['density_grid']).requires_grad_(True)
            if irregular_shape:
                o_chunks, d_chunks = o_img.split(10000), d_img.split(10000)
            else:
                o_chunks = o_img[::downrate, ::downrate].to(dev).flatten(0, -2).split(10000)
                d_chunks = d_img[::downrate, ::downrate].to(dev).flatten(0, -2).split(10000)
            for o, d in zip(o_chunks, d_chunks):
                vec = torch.where(d == 0, torch.full_like(d, 1e-6), d)
                rate_a = (s['xyz_max'] - o) / vec
                rate_b = (s['xyz_min'] - o) / vec
                t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near, max=far)
                step = stepsize * s['voxel_size'] * rng
                interpx = t_min[..., None] + step / d.norm(dim=-1, keepdim=True)
                pts = o[..., None, :] + d[..., None, :] * interpx[..., None]
                self.grad_query(ones, pts, s['xyz_min'], s['xyz_max'], 0).sum().backward()
            with torch.no_grad():
                count += (ones.grad > 1)
        return count


@torch.no_grad()
def get_training_rays_in_maskcache_sampling(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, model,
                                            render_kwargs, get_rays=None):
    """Keep only the pixels whose rays hit the coarse geometry (dvgo.py:619-657); returns the flattened
    (rgb, rays_o, rays_d, viewdirs, imsz).  model: DirectVoxGORenderer (anything with hit_coarse_geo)."""
    assert len(rgb_tr_ori) == len(train_poses) and len(rgb_tr_ori) == len(Ks) and len(rgb_tr_ori) == len(HW)
    if ndc:
        raise NotImplementedError("NDC rays belong to the DirectMPIGO path (out of scope, SURVEY.md section 2)")
    if get_rays is None:
        from .fourier_render import get_rays_of_a_view as get_rays
    dev = rgb_tr_ori[0].device
    total = sum(im.shape[0] * im.shape[1] for im in rgb_tr_ori)
    rgb_tr = torch.zeros(total, 3, device=dev)
    rays_o_tr, rays_d_tr, viewdirs_tr = torch.zeros_like(rgb_tr), torch.zeros_like(rgb_tr), torch.zeros_like(rgb_tr)
    imsz, top = [], 0
    for c2w, img, (H, W), K in zip(train_poses, rgb_tr_ori, HW, Ks):
        assert img.shape[:2] == (H, W)
        rays_o, rays_d, viewdirs = get_rays(H, W, K, torch.as_tensor(c2w, dtype=torch.float32).to(dev),
                                            inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
        mask = model.hit_coarse_geo(rays_o=rays_o, rays_d=rays_d, **render_kwargs).to(dev)   # whole image at once
        n = int(mask.sum())
        rgb_tr[top:top + n] = img[mask]
        rays_o_tr[top:top + n] = rays_o[mask]
        rays_d_tr[top:top + n] = rays_d[mask]
        viewdirs_tr[top:top + n] = viewdirs[mask]
        imsz.append(n)
        top += n
    return rgb_tr[:top], rays_o_tr[:top], rays_d_tr[:top], viewdirs_tr[:top], imsz
