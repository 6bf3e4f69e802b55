import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, numpy as np, synth
import test_gpu_voxgo_train as T
from unboundednerfpytorch_amd import grid as G
dev = torch.device("cuda", 0)
for case in synth.DCVGO_CASES:
    m, name, (o, d, v), kw, R, seed = T.build("dcvgo", case, dev)
    with torch.no_grad():
        m.fused_forward = True
        a = m(o, d, v, global_step=1, is_train=True, **kw)
        m.fused_forward = False
        b = m(o, d, v, global_step=1, is_train=True, **kw)
    print(name, "n", a["weights"].numel(), b["weights"].numel())
    for k in ("ray_id", "step_id", "t", "raw_density", "raw_alpha", "weights", "alphainv_last"):
        if a[k].shape == b[k].shape:
            neq = (a[k] != b[k])
            print("  ", k, "mismatches", int(neq.sum()), "max abs", float((a[k].float() - b[k].float()).abs().max()))
    # where do the densities differ: inner or contracted samples?
    pts, inner, t = m.sample_ray(ori_rays_o=o, ori_rays_d=d, **kw)
    inner_of = inner[b["ray_id"], b["step_id"]]
    neq = a["raw_density"] != b["raw_density"]
    print("   density mismatches: inner", int((neq & inner_of).sum()), "of", int(inner_of.sum()), " contracted", int((neq & ~inner_of).sum()), "of", int((~inner_of).sum()))
    # the points of the fused op vs the torch chain
    hc = m._host_consts()
    tt = m.sample_table(kw["stepsize"], dev)
    cfg = {'mode': 'dcvgo', 'act_shift': hc['act_shift'], 'interval': m._step_consts(kw["stepsize"])[0], 'thres': float(m.fast_color_thres),
           'mask_scale': hc['mask_scale'], 'mask_shift': hc['mask_shift'], 'scene_center': hc['scene_center'], 'scene_radius': hc['scene_radius'],
           'bg_len': m.bg_len, 'norm_l2': m.contracted_norm == 'l2', 'dist_thres': (2 + 2 * m.bg_len) / m.world_len * kw["stepsize"] * 0.95}
    with torch.no_grad():
        p2, dens2, al2, w2, ainv, ray2, step2, t2, in2 = G.TrainSampleVox.apply(m.density.grid, o.contiguous(), d.contiguous(), tt, m.xyz_min, m.xyz_max, m.mask_cache.mask, cfg)
    ref_p = pts[ray2, step2]
    dp = (p2 - ref_p).abs().amax(-1)
    print("   points: mismatching", int((dp > 0).sum()), "of", dp.numel(), "max", float(dp.max()), " among inner", int(((dp > 0) & in2).sum()), " inner flag equal", bool(torch.equal(in2, inner[ray2, step2])))
    # pieces of the torch chain
    oo = (o - m.scene_center) / m.scene_radius
    dd = d / d.norm(dim=-1, keepdim=True)
    raw = oo[:, None, :] + dd[:, None, :] * tt[None, :, None]
    nrm = raw.abs().amax(dim=-1, keepdim=True) if m.contracted_norm == 'inf' else raw.norm(dim=-1, keepdim=True)
    v1 = raw / nrm * ((1 + m.bg_len) - m.bg_len / nrm)
    v2 = raw / nrm * (torch.tensor(1 + m.bg_len, device=dev) - torch.tensor(m.bg_len, device=dev) / nrm)
    v3 = raw / nrm * (torch.tensor(1 + m.bg_len, device=dev) - nrm.reciprocal() * torch.tensor(m.bg_len, device=dev))
    sel = ~in2
    for nm, vv in (("python-number form", v1), ("tensor division", v2), ("reciprocal * bg", v3)):
        q = vv[ray2, step2]
        print("   kernel vs", nm, ": mismatching contracted points", int(((p2 - q).abs().amax(-1) > 0)[sel].sum()), "of", int(sel.sum()))
