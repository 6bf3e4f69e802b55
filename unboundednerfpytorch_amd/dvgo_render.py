"""DirectVoxGORenderer: inference forward of the reference's bounded model dvgo.DirectVoxGO
(/root/reference/FourierGrid/dvgo.py:306-425, BASELINE.json configs[0]) composed from the drop-in kernels:
sample_pts_on_rays (ray-AABB clip + variable-length marching) -> maskcache_lookup -> dense grid query ->
raw2alpha -> alpha2weight -> k0 query -> rgbnet -> per-ray sums.  Same call signature and return keys as the
reference forward (`rgb_marched`, `depth`, `alphainv_last`, `weights`, `raw_alpha`, `raw_rgb`, `ray_id`).

The compaction steps (boolean masks) and the tiny rgbnet use torch on the device, exactly like the
reference does; every kernel the reference has natively is the HIP one.
"""
import torch
import torch.nn.functional as F

from . import render_utils_cuda
from .grid import grid_query


class DirectVoxGORenderer:
    """state: xyz_min/xyz_max [3], density_grid [1,1,X,Y,Z], k0_grid [1,C,X,Y,Z], rgbnet_weights/biases (lists),
    mask [mx,my,mz] bool, xyz2ijk_scale/shift [3], act_shift, voxel_size, voxel_size_ratio (0-d tensors or floats),
    fast_color_thres, rgbnet_direct, viewbase_pe."""

    def __init__(self, state, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("DirectVoxGORenderer needs a HIP device (no CPU path)")
        self.device = dev
        self.s = {k: (v.to(dev).contiguous() if torch.is_tensor(v) else
                      ([x.to(dev).contiguous() for x in v] if isinstance(v, list) else v)) for k, v in state.items()}
        self.viewfreq = torch.tensor([float(2 ** i) for i in range(int(state["viewbase_pe"]))], device=dev)

    @torch.no_grad()
    def forward(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        s = self.s
        assert rays_o.dim() == 2 and rays_o.shape[-1] == 3, 'Only suuport point queries in [N, 3] format'
        N = rays_o.shape[0]
        stepsize = render_kwargs['stepsize']
        far = 1e9  # the given far can be too small while rays stop when hitting scene bbox (dvgo.py:318)
        stepdist = stepsize * s['voxel_size']
        ray_pts, mask_outbbox, ray_id, step_id = render_utils_cuda.sample_pts_on_rays(
            rays_o.contiguous(), rays_d.contiguous(), s['xyz_min'], s['xyz_max'], render_kwargs['near'], far, stepdist)[:4]
        inb = ~mask_outbbox
        ray_pts, ray_id, step_id = ray_pts[inb], ray_id[inb], step_id[inb]
        interval = stepsize * s['voxel_size_ratio']
        m = render_utils_cuda.maskcache_lookup(s['mask'], ray_pts.contiguous(), s['xyz2ijk_scale'], s['xyz2ijk_shift'])
        ray_pts, ray_id, step_id = ray_pts[m], ray_id[m], step_id[m]
        density = grid_query(s['density_grid'], ray_pts, s['xyz_min'], s['xyz_max'], 0)
        alpha = render_utils_cuda.raw2alpha(density.flatten().contiguous(), s['act_shift'], interval)[1]
        thres = float(s['fast_color_thres'])
        if thres > 0:
            k = alpha > thres
            ray_pts, ray_id, step_id, alpha = ray_pts[k], ray_id[k], step_id[k], alpha[k]
        weights, _, alphainv_last = render_utils_cuda.alpha2weight(alpha.contiguous(), ray_id.contiguous(), N)[:3]
        if thres > 0:
            k = weights > thres
            weights, alpha, ray_pts, ray_id, step_id = weights[k], alpha[k], ray_pts[k], ray_id[k], step_id[k]
        k0 = grid_query(s['k0_grid'], ray_pts, s['xyz_min'], s['xyz_max'], 0)
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
        rgb_marched = torch.zeros(N, 3, device=self.device).index_add_(0, ray_id, weights.unsqueeze(-1) * rgb)
        rgb_marched += alphainv_last.unsqueeze(-1) * render_kwargs['bg']
        out = {'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha,
               'raw_rgb': rgb, 'ray_id': ray_id}
        if render_kwargs.get('render_depth', False):
            out['depth'] = torch.zeros(N, device=self.device).index_add_(0, ray_id, weights * step_id)
        return out

    __call__ = forward
