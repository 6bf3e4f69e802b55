"""BASELINE.json configs[1] as worded ("dcvgo contracted bg grid"): one 1920x1080 frame through the FUSED
DirectContractedVoxGO inference path (dcvgo_render.DirectContractedVoxGORenderer.render_rays -> ugrid_render_march_dcvgo +
ugrid_render_shade, F = 0) at the model shape of configs/nerf_unbounded/nerf_unbounded_default.py:34-52 (+ default.py:
119-121): num_voxels = 320^3 single-level density and 12-channel feature grids, rgbnet 39-128-128-3, stepsize 0.5
(S = 1068 samples per ray), fast_color_thres 1e-4, mask cache from the occupancy field.

    python tools/bench_dcvgo.py [--grid 320] [--steps 5] [--out file.json]                       (GPU box)

Prints one JSON line: ms per frame (march / shade split), Msamples/s, the fraction of samples the cumdist rule and the mask
cache let through, and the per-ray agreement with the composed forward (drop-in kernels + torch glue, the round-2 path)
on 4 x 8192 rays of the same frame, with that path's own time per ray beside it."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_dcvgo_state(G, device, seed=0):
    """trained-like single-level grids: the occupancy field of bench.make_state_surfaces (level 0 carries 7x the target
    density there), low-pass feature noise, nn.Linear-initialised rgbnet, mask cache = dilated (alpha > 1e-5)"""
    import math
    import bench
    from unboundednerfpytorch_amd.dcvgo_render import dcvgo_state_from_params
    st = bench.make_state_surfaces(G, device, seed=seed)
    dens = (st["density_grid"][0:1] / 7.0).contiguous()            # [1,1,G,G,G]
    k0 = st["k0_grid"][0:1].contiguous()                           # [1,12,G,G,G]
    alpha = 1 - (1 + torch.exp(dens + math.log(1 / (1 - 1e-4) - 1))) ** (-0.5)
    mask = torch.nn.functional.max_pool3d(alpha, kernel_size=3, padding=1, stride=1)[0, 0] > 1e-5
    out = dcvgo_state_from_params([-1, -1, -1], [1, 1, 1], G ** 3, G ** 3, 1e-4, dens, k0, st["rgbnet_weights"], st["rgbnet_biases"],
                                  mask, 1e-4, contracted_norm="inf", viewbase_pe=4)
    assert int(out["world_len"]) == G, (out["world_len"], G)
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
    ap.add_argument("--grid", type=int, default=320)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--out", default=None)
    ap.add_argument("--burn-streams", type=int, default=0, help="diagnostic: take this many pool streams first (profiles/r06/side_stream_queues.txt)")
    args = ap.parse_args(argv)
    import bench
    from unboundednerfpytorch_amd.dcvgo_render import DirectContractedVoxGORenderer
    from unboundednerfpytorch_amd.fourier_render import get_rays_of_pixel_index, pixel_tile_order, untile
    dev = torch.device("cuda", 0)
    for _ in range(int(getattr(args, "burn_streams", 0))):
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            torch.zeros(1, device=dev)
    G, H, W = args.grid, args.height, args.width
    state = make_dcvgo_state(G, dev)
    rend = DirectContractedVoxGORenderer(state, dev)
    assert rend.fused_supported()
    K = [[1600.0 * W / 1920.0, 0, W / 2.0], [0, 1600.0 * W / 1920.0, H / 2.0], [0, 0, 1]]
    c2w = bench.camera(0, dev)
    order = pixel_tile_order(H, W, dev)
    bg = torch.ones(3, device=dev)

    def frame(timing=None):
        ro, rd, vd = get_rays_of_pixel_index(H, W, K, c2w, order)
        out = rend.render_rays(ro, rd, vd, stepsize=0.5, bg=bg, render_depth=True, ray_order="coherent", timing=timing)
        return {k: untile(v, H, W) for k, v in out.items()}, (ro, rd, vd)

    for _ in range(args.warmup):
        out, rays = frame()
    torch.cuda.synchronize()
    timing = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, rays = frame(timing)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    dt2 = two_in_flight(rend, frame, args.steps, dev)
    dtn = {n: two_in_flight(rend, frame, args.steps, dev, n) * 1e3 for n in (3, 4)}
    fr = rend._fused
    S = fr.tables(0.5)[2]
    M = fr.survivors_of_last_chunk()
    R = H * W
    march = sum(ev[0].elapsed_time(ev[1]) for ev, _ in timing) / args.steps
    shade = sum(ev[-2].elapsed_time(ev[-1]) for ev, _ in timing) / args.steps
    # agreement with (and time of) the composed forward on a sample of the frame's rays
    ro, rd, vd = rays
    starts = [int(i * (R - 8192) / 3) // 64 * 64 for i in range(4)]
    worst = {k: 0.0 for k in ("rgb_marched", "depth", "alphainv_last", "wsum_mid")}
    n_bad, kept = 0, 0
    t_comp = 0.0
    for b in starts:
        o_, d_, v_ = ro[b:b + 8192].contiguous(), rd[b:b + 8192].contiguous(), vd[b:b + 8192].contiguous()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ref = rend(o_, d_, v_, stepsize=0.5, bg=bg, render_depth=True)
        torch.cuda.synchronize()
        t_comp += time.perf_counter() - t1
        got = rend.render_rays(o_, d_, v_, stepsize=0.5, bg=bg, render_depth=True, ray_order="coherent")
        kept += int(ref["weights"].numel())
        bad = torch.zeros(8192, dtype=torch.bool, device=dev)
        for k in worst:
            e = (got[k] - ref[k]).abs()
            e = e.amax(dim=1) if e.dim() == 2 else e
            worst[k] = max(worst[k], float(e.max()))
            bad |= e > 1e-4
        n_bad += int(bad.sum())
    line = json.dumps({
        "workload": "DirectContractedVoxGO render, %dx%d rays x S=%d samples, G=%d^3 single-level grids, C=12, rgbnet 39-128-128-3, "
                    "stepsize 0.5, thres 1e-4, mask cache, trained-like synthetic fields (tools/bench_dcvgo.make_dcvgo_state)" % (W, H, S, G),
        "path": "fused: ugrid_render_march_dcvgo + ugrid_render_shade (F = 0)", "ms_per_frame": dt * 1e3, "ms_per_frame_two_in_flight": dt2 * 1e3, "ms_n_in_flight": dtn,
        "kernels_ms": {"march_dcvgo": march, "shade": shade}, "value": R * S / dt / 1e6, "unit": "Msamples/s", "rays_per_sec": R / dt,
        "survivors_M": M, "survivor_frac": M / float(R * S), "terminated_ray_frac": float((out["alphainv_last"] < 1e-3).float().mean()),
        "vs_composed_forward": {"rays": 4 * 8192, "linf": worst, "rays_above_1e-4": n_bad,
                                "composed_us_per_ray": t_comp / (4 * 8192) * 1e6, "fused_us_per_ray": dt / R * 1e6,
                                "composed_survivors": kept},
        "finite": bool(torch.isfinite(out["rgb_marched"]).all())})
    if not quiet:
        print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(line + "\n")
    return json.loads(line)


if __name__ == "__main__":
    main()
