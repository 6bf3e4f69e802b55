"""fp64 evaluation of FourierGridModel.forward for a handful of rays (tool for tools/gpu_parity_probe.py): same
pipeline and thresholds as oracle/model_oracle.fouriergrid_render, every tensor in double.  Tells whether a
GPU-vs-oracle difference on a ray is the GPU's error or fp32 sensitivity of the ray itself (oracle32 vs fp64)."""
import torch
from oracle import model_oracle as mo


def render_fp64(state, rays_o, rays_d, viewdirs, stepsize, grids64=None):
    D = torch.float64
    F_num, thres = int(state['fourier_freq_num']), float(state['fast_color_thres'])
    t = mo.sample_t(int(state['world_len']), stepsize, float(state['bg_len'])).to(D)
    pts, _ = mo.contracted_sample_ray(rays_o.to(D), rays_d.to(D), state['scene_center'].to(D), state['scene_radius'].to(D),
                                      t, float(state['bg_len']), state.get('contracted_norm', 'inf'))
    interval = float(state['voxel_size_ratio']) * stepsize
    lo, hi = state['xyz_min'].to(D), state['xyz_max'].to(D)
    dg, kg = grids64 if grids64 is not None else (state['density_grid'].to(D), state['k0_grid'].to(D))
    dens = mo.fourier_grid_query(dg, pts, lo, hi, F_num)                       # [R,S]
    alpha = 1 - (1 + torch.exp(dens + float(state['act_shift']))) ** (-interval)
    R, S = alpha.shape
    out = {'rgb_marched': torch.zeros(R, 3, dtype=D), 'depth': torch.zeros(R, dtype=D), 'alphainv_last': torch.ones(R, dtype=D)}
    s_tab = 1 - 1 / (1 + t)
    emb = mo.viewdir_embedding(viewdirs.to(D), int(state['viewbase_pe'])) if len(state['rgbnet_weights']) else None
    for r in range(R):
        T = 1.0
        keep, ws = [], []
        for j in range(S):
            a = float(alpha[r, j])
            if not a > thres:
                continue
            w = T * a
            T = T * (1 - a)
            if w > thres:
                keep.append(j); ws.append(w)
            if T < 1e-3:
                break
        out['alphainv_last'][r] = T
        if keep:
            idx = torch.tensor(keep)
            wt = torch.tensor(ws, dtype=D)
            k0 = mo.fourier_grid_query(kg, pts[r, idx], lo, hi, F_num if kg.shape[0] > 1 else 0)
            if emb is not None:
                x = torch.cat([k0, emb[r][None].expand(len(keep), -1)], -1)
                rgb = torch.sigmoid(mo.rgbnet_apply([w_.to(D) for w_ in state['rgbnet_weights']],
                                                    [b_.to(D) for b_ in state['rgbnet_biases']], x))
            else:
                rgb = torch.sigmoid(k0)
            out['rgb_marched'][r] = (wt[:, None] * rgb).sum(0)
            out['depth'][r] = (wt * s_tab[idx]).sum()
    return out
