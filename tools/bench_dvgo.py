"""BASELINE.json configs[0] (lego-shaped bounded DVGO) at its real size: one 800x800 view through the FUSED DirectVoxGO
inference path (dvgo_render.DirectVoxGORenderer.render_rays -> ugrid_render_march_dvgo + ugrid_render_shade, F = 0) at the
model shape of configs/nerf/lego.py + default.py:77-121 -- fine stage: num_voxels = 160^3 dense density and 12-channel
feature grids in the lego box, rgbnet_direct 39-128-128-3, stepsize 0.5, near / far 2 / 6, white background, fast_color_thres
1e-4, mask cache from the occupancy field.

    python tools/bench_dvgo.py [--grid 160] [--steps 5] [--out file.json]                       (GPU box)

Prints one JSON line: ms per view (march / shade split), rays/s, steps marched, and the per-ray agreement with the composed
forward (drop-in kernels + torch glue incl. the count -> cumsum -> host read -> fill sampling, the reference's own structure)
on the same rays with that path's time beside it."""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LEGO_MIN, LEGO_MAX = [-0.67, -1.2, -0.37], [0.67, 1.2, 1.03]


def make_dvgo_state(G, device, seed=0):
    """trained-like dense grids in the lego box: an object of soft-edged spheres and a base plate in the middle of the box
    (raw density -6 in empty space, +16 inside, 1.5-voxel transitions: solids saturate in 1-2 steps, like a converged DVGO
    model), low-pass feature noise, nn.Linear-initialised rgbnet, mask cache = dilated (alpha > 1e-5)"""
    from unboundednerfpytorch_amd.dvgo_render import dvgo_state_from_params
    g = torch.Generator(device=device)
    g.manual_seed(seed + 2000)
    lo, hi = torch.Tensor(LEGO_MIN), torch.Tensor(LEGO_MAX)
    vs = float(((hi - lo).prod() / G ** 3).pow(1 / 3))
    ws = ((hi - lo) / vs).long().tolist()
    X, Y, Z = torch.meshgrid(*[torch.linspace(LEGO_MIN[i], LEGO_MAX[i], ws[i], device=device) for i in range(3)], indexing="ij")
    w = 1.5 * vs
    occ = torch.sigmoid((0.06 - (Z + 0.2).abs()) / w) * torch.sigmoid((0.45 - X.abs()) / w) * torch.sigmoid((0.9 - Y.abs()) / w)  # plate
    cg = torch.Generator()
    cg.manual_seed(seed + 11)
    for _ in range(12):
        c = (torch.rand(3, generator=cg) * torch.tensor([0.7, 1.5, 0.7]) + torch.tensor([-0.35, -0.75, -0.1])).tolist()
        r = float(torch.rand(1, generator=cg) * 0.15 + 0.07)
        occ = torch.maximum(occ, torch.sigmoid((r - ((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2).sqrt()) / w))
    dens = (-6.0 + 22.0 * occ)[None, None].contiguous()
    k0 = torch.empty(1, 12, *ws, device=device).normal_(0.0, 1.0, generator=g)
    for _ in range(2):
        k0 = torch.nn.functional.avg_pool3d(k0, 5, stride=1, padding=2, count_include_pad=False)
    k0 = (k0 / k0.std() * 0.5).contiguous()
    torch.manual_seed(seed + 5)
    net = [torch.nn.Linear(39, 128), torch.nn.Linear(128, 128), torch.nn.Linear(128, 3)]
    torch.nn.init.constant_(net[-1].bias, 0)
    alpha = 1 - (1 + torch.exp(dens + math.log(1 / (1 - 1e-2) - 1))) ** (-0.5)
    mask = torch.nn.functional.max_pool3d(alpha, kernel_size=3, padding=1, stride=1)[0, 0] > 1e-5
    out = dvgo_state_from_params(LEGO_MIN, LEGO_MAX, G ** 3, G ** 3, 1e-2, dens, k0, [m.weight.detach() for m in net],
                                 [m.bias.detach() for m in net], mask, 1e-4, True, viewbase_pe=4)
    assert out["world_size"].tolist() == ws
    return out


def two_in_flight(rend, frame, steps, dev, n=2):
    """seconds per frame with consecutive frames alternating between two streams / two work lists (run_render.render_viewpoints' default)"""
    pair = [torch.cuda.Stream(dev) for _ in range(n)]
    for st in pair:
        st.wait_stream(torch.cuda.current_stream(dev))
    best = None
    for rep in range(2):                  # the first batch warms the second work list
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2 * steps):
            rend.use_workspace_slot(i % n)
            with torch.cuda.stream(pair[i % n]):
                frame()
        torch.cuda.synchronize()
        best = (time.perf_counter() - t0) / (2 * steps)
    rend.use_workspace_slot(0)
    return best


def main(argv=None, quiet=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=160)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--out", default=None)
    ap.add_argument("--burn-streams", type=int, default=0, help="diagnostic: take this many pool streams first (profiles/r06/side_stream_queues.txt)")
    args = ap.parse_args(argv)
    from unboundednerfpytorch_amd.dvgo_render import DirectVoxGORenderer
    from unboundednerfpytorch_amd.fourier_render import get_rays_of_pixel_index, pixel_tile_order, untile
    dev = torch.device("cuda", 0)
    for _ in range(int(getattr(args, "burn_streams", 0))):
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            torch.zeros(1, device=dev)
    G, H, W = args.grid, args.height, args.width
    rend = DirectVoxGORenderer(make_dvgo_state(G, dev), dev)
    assert rend.fused_supported()
    f = 1111.1 * W / 800.0                                      # the blender cameras: 800 px, camera_angle_x = 0.6911
    K = [[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]]
    c2w = torch.tensor([[-0.9999, 0.0042, -0.0133, -0.0538], [-0.0140, -0.2997, 0.9539, 3.8455], [0.0, 0.9540, 0.2997, 1.2081]],
                       device=dev)
    order = pixel_tile_order(H, W, dev)
    kw = dict(near=2.0, far=6.0, stepsize=0.5, bg=1, render_depth=True)

    def view(timing=None):
        ro, rd, vd = get_rays_of_pixel_index(H, W, K, c2w, order)
        out = rend.render_rays(ro, rd, vd, ray_order="coherent", timing=timing, **kw)
        return {k: untile(v, H, W) for k, v in out.items()}, (ro, rd, vd)

    for _ in range(args.warmup):
        out, rays = view()
    torch.cuda.synchronize()
    timing = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, rays = view(timing)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    dt2 = two_in_flight(rend, view, args.steps, dev)
    dtn = {n: two_in_flight(rend, view, args.steps, dev, n) * 1e3 for n in (3, 4)}
    fr = rend._fused
    M = fr.survivors_of_last_chunk()
    R = H * W
    march = sum(ev[0].elapsed_time(ev[1]) for ev, _ in timing) / args.steps
    shade = sum(ev[-2].elapsed_time(ev[-1]) for ev, _ in timing) / args.steps
    # the composed forward on the same rays, the reference's 8192-ray render chunks (run_render.py:52-58)
    ro, rd, vd = rays
    worst = {k: 0.0 for k in ("rgb_marched", "depth", "alphainv_last")}
    got = rend.render_rays(ro, rd, vd, ray_order="coherent", **kw)
    n_bad, kept, n_steps = 0, 0, 0
    for rep in range(2):                 # second pass timed (allocator warm)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for b in range(0, R, 8192):
            ref = rend(ro[b:b + 8192], rd[b:b + 8192], vd[b:b + 8192], **kw)
            if rep == 1:
                kept += int(ref["weights"].numel())
                bad = torch.zeros(ref["rgb_marched"].shape[0], dtype=torch.bool, device=dev)
                for k, tol in (("rgb_marched", 1e-4), ("alphainv_last", 1e-4), ("depth", 1e-2)):
                    e = (got[k][b:b + 8192] - ref[k]).abs()
                    e = e.amax(dim=1) if e.dim() == 2 else e
                    worst[k] = max(worst[k], float(e.max()))
                    bad |= e > tol
                n_bad += int(bad.sum())
        torch.cuda.synchronize()
        t_comp = time.perf_counter() - t1
    from unboundednerfpytorch_amd import render_utils_cuda as ru
    s = rend.s
    n_steps = int(ru.sample_pts_on_rays(ro, rd, s["xyz_min"], s["xyz_max"], 2.0, 1e9, 0.5 * s["voxel_size"])[2].numel())
    line = json.dumps({
        "workload": "DirectVoxGO render (configs[0] shape), %dx%d rays, lego box, G=%d^3 -> world size %s, C=12, rgbnet_direct "
                    "39-128-128-3, stepsize 0.5, near 2, thres 1e-4, mask cache, trained-like synthetic fields "
                    "(tools/bench_dvgo.make_dvgo_state)" % (W, H, G, s["world_size"].tolist()),
        "path": "fused: ugrid_render_march_dvgo + ugrid_render_shade (F = 0)", "ms_per_view": dt * 1e3, "ms_per_view_two_in_flight": dt2 * 1e3, "ms_n_in_flight": dtn,
        "kernels_ms": {"march_dvgo": march, "shade": shade}, "rays_per_sec": R / dt, "steps_marched_M": n_steps / 1e6,
        "value": n_steps / dt / 1e6, "unit": "Msamples/s", "survivors_M": M / 1e6,
        "terminated_ray_frac": float((out["alphainv_last"] < 1e-3).float().mean()),
        "hit_ray_frac": float((out["alphainv_last"] < 0.99).float().mean()),
        "vs_composed_forward": {"rays": R, "linf": worst, "rays_outside_tol(1e-4 rgb/alphainv, 1e-2 depth)": n_bad,
                                "composed_ms_per_view": t_comp * 1e3, "speedup": t_comp / dt, "composed_survivors": kept},
        "finite": bool(torch.isfinite(out["rgb_marched"]).all())})
    if not quiet:
        print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(line + "\n")
    return json.loads(line)


if __name__ == "__main__":
    main()
