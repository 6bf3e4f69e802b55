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


KEYS = ("rgb_marched", "depth", "alphainv_last")


def _err(a, b):
    e = (a.double() - b.double()).abs()
    return e.amax(dim=1) if e.dim() == 2 else e


def ground_truth_study(cpu_state, rays_cpu, evaluations, stepsize, n_random=1024, seed=0, slack=2e-5, bound=1e-4):
    """replace "the reference disagrees with itself by that much" with a GROUND TRUTH.

    `evaluations`: {"fused": {...}, "ref_cpu": {...}, "ref_gpu": {...} (optional)} -- fp32 outputs (rgb_marched [n,3], depth [n],
    alphainv_last [n]) of the same n rays `rays_cpu` = (rays_o, rays_d, viewdirs).  The fp64 evaluation (render_fp64: the
    same formula, thresholds and sample table, every tensor and every libm call in double) is run on
      * every ray where the fused render is further than `bound` from ANY fp32 reference in ANY output, and
      * `n_random` rays drawn uniformly (seeded),
    and for each selected ray and output the distances |x - fp64| of the evaluations are compared.  Returns a dict with, per
    output: L-inf / mean / rays above `bound` of each evaluation against fp64, and per fp32 reference `name`
      violations[name] = rays where |fused - fp64| > |name - fp64| + slack        (the fused render is further from the truth
                                                                                   than that reference by more than the slack)
      bound_where_ref_ok[name] = L-inf of |fused - fp64| over the rays where |name - fp64| <= bound in all outputs
    """
    import torch
    ro, rd, vd = rays_cpu
    n = ro.shape[0]
    fused = evaluations["fused"]
    refs = {k: v for k, v in evaluations.items() if k != "fused" and v is not None}
    far = torch.zeros(n, dtype=torch.bool)
    for r in refs.values():
        for k in KEYS:
            far |= _err(fused[k], r[k]) > bound
    g = torch.Generator().manual_seed(seed)
    pick = torch.zeros(n, dtype=torch.bool)
    pick[torch.randperm(n, generator=g)[:min(n_random, n)]] = True
    sel = torch.nonzero(far | pick).flatten()
    grids64 = (cpu_state["density_grid"].double(), cpu_state["k0_grid"].double())
    truth = render_fp64(cpu_state, ro[sel], rd[sel], vd[sel], stepsize, grids64=grids64)
    del grids64
    res = {"rays_total": int(n), "rays_evaluated_in_fp64": int(sel.numel()), "rays_selected_because_fused_is_far_from_a_reference": int(far.sum()),
           "rays_random": int(pick.sum()), "slack": slack, "bound": bound, "distance_to_fp64": {}, "violations": {}, "bound_where_ref_ok": {}}
    dist = {name: {k: _err(ev[k][sel], truth[k]) for k in KEYS} for name, ev in [("fused", fused)] + list(refs.items())}
    for name, d in dist.items():
        res["distance_to_fp64"][name] = {k: {"linf": float(e.max()), "mean_abs": float(e.mean()), "rays_above_bound": int((e > bound).sum())}
                                         for k, e in d.items()}
    sound = torch.ones(sel.numel(), dtype=torch.bool)          # rays on which EVERY fp32 reference is within bound / 2 of the truth
    for name in refs:
        viol, ok_all = {}, torch.ones(sel.numel(), dtype=torch.bool)
        for k in KEYS:
            v = dist["fused"][k] > dist[name][k] + slack
            rv = dist[name][k] > dist["fused"][k] + slack
            viol[k] = {"rays": int(v.sum()), "worst_excess": float((dist["fused"][k] - dist[name][k]).max()),
                       "rays_where_the_reference_is_further_than_fused": int(rv.sum()),
                       "worst_excess_of_the_reference": float((dist[name][k] - dist["fused"][k]).max())}
            ok_all &= dist[name][k] <= bound
            sound &= dist[name][k] <= 0.5 * bound
        res["violations"][name] = viol
        res["bound_where_ref_ok"][name] = {"rays_where_ref_within_bound_of_fp64": int(ok_all.sum()),
                                           **{k: (float(dist["fused"][k][ok_all].max()) if bool(ok_all.any()) else None) for k in KEYS}}
    res["fused_on_rays_where_all_references_are_within_half_bound"] = {
        "rays": int(sound.sum()), **{k: (float(dist["fused"][k][sound].max()) if bool(sound.any()) else None) for k in KEYS}}
    # per-ray distances (for offline analysis; 3 x n_eval floats per evaluation)
    res["per_ray"] = {"selected_because_far": far[sel].tolist(),
                      **{name: {k: [float("%.4g" % x) for x in d[k].tolist()] for k in KEYS} for name, d in dist.items()}}
    return res
