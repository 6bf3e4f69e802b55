"""S3 (BASELINE.json configs[2]): one Tanks&Temples-'truck'-shaped training iteration at the real
sizes -- P = 9 Fourier levels (F = 4), G = 200^3, C = 12, N_rand = 4096 random rays x S = 668 samples (stepsize 0.5),
loss terms / TV window / optimizer of configs/tankstemple_unbounded/truck_single.py:56-82 -- through
fourier_model.FourierGridModel + train_step.train_iteration (the loop body of run_train.py:185-296).

    python tools/bench_train_step.py [--steps 10] [--grid 200] [--fused 0|1]      (GPU box)

Prints one JSON line: ms per step split by phase (HIP events), survivors, the k0-sized streaming passes in GB/s."""
import argparse
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TRUCK_CFG = dict(  # fine_train of truck_single.py (+ default.py)
    N_rand=4096, weight_main=1.0, weight_freq=0.0, weight_entropy_last=1e-3, weight_rgbper=1e-2, weight_nearclip=0.0,
    weight_distortion=0.01, weight_tv_density=1e-6, weight_tv_k0=1e-7, tv_before=1e9, tv_dense_before=10000, tv_after=0,
    tv_every=1, lrate_density=1e-1, lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_decay=20, skip_zero_grad_fields=['density', 'k0'],
    pg_scale=[])


def make_model(G, F, device, fused, channels_last=True):
    import bench
    from unboundednerfpytorch_amd.fourier_model import FourierGridModel
    m = FourierGridModel(xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1], num_voxels_density=G ** 3, num_voxels_base_density=G ** 3,
                         num_voxels_rgb=G ** 3, num_voxels_base_rgb=G ** 3, num_voxels_viewdir=-1, alpha_init=1e-4,
                         fast_color_thres=1e-4, fourier_freq_num=F, rgbnet_dim=12, channels_last_grids=bool(channels_last)).to(device)
    if hasattr(m, "fused_forward"):
        m.fused_forward = bool(fused)
    # trained-like fields (bench.make_state_surfaces is the F = 3 version of the same recipe)
    st = bench.make_state_surfaces(G, device, seed=0)
    P = 1 + 2 * F
    with torch.no_grad():
        g = torch.Generator(device=device)
        g.manual_seed(5)
        m.density.grid.normal_(0.0, 0.3, generator=g)
        m.density.grid[0, 0] = st["density_grid"][0, 0] * (P / 7.0)          # the level mean restores the occupancy field
        m.k0.grid.normal_(0.0, 0.5, generator=g)
        m.k0.grid[:7] += st["k0_grid"]
    del st
    torch.cuda.empty_cache()
    return m


def random_rays(n, device, seed):
    """N_rand rays of random cameras on a ring around the scene looking inwards with random pixel offsets."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ang = torch.rand(n, device=device, generator=g) * 2 * math.pi
    o = torch.stack([0.55 * torch.cos(ang), 0.55 * torch.sin(ang), 0.25 + 0.3 * torch.rand(n, device=device, generator=g)], -1)
    tgt = (torch.rand(n, 3, device=device, generator=g) - 0.5) * torch.tensor([1.6, 1.6, 1.2], device=device)
    d = tgt - o
    d = d * (0.5 + torch.rand(n, 1, device=device, generator=g))
    v = d / d.norm(dim=-1, keepdim=True)
    rgb = torch.rand(n, 3, device=device, generator=g)
    return o.contiguous(), d.contiguous(), v.contiguous(), rgb


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--burn-streams", type=int, default=0, help="diagnostic: take this many streams from torch's pool first (which pool stream, "
                    "hence which hardware queue, the step's side stream lands on depends on how many were handed out before)")
    ap.add_argument("--blocks", type=int, default=1, help="clock the timed steps in B equal blocks and report the median block (steps must be a multiple)")
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--freq", type=int, default=4)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--first-step", type=int, default=1, help="global_step of the first iteration: 1 = the dense-TV phase of "
                    "truck_single (global_step < tv_dense_before = 10000), 10001 = the masked-TV phase that follows")
    ap.add_argument("--overlap", type=int, default=1, help="k0 TV + Adam pass on a second stream (train_iteration overlap_k0_update)")
    ap.add_argument("--fused-loss", type=int, default=1, help="compositing + loss as one op (ops.RenderLoss) or the torch chain")
    ap.add_argument("--touch", type=int, default=1, help="touched-line bitmap of the recycled k0 gradient (_gradpool.touch_enabled): "
                    "the masked TV / Adam passes visit only the lines the backward marked; 0 = the scanning kernels")
    ap.add_argument("--lazy-loss", type=int, default=0, help="train_iteration(return_tensors=True): no host read of loss / psnr per step")
    ap.add_argument("--tune", action="append", default=[], help="key=value for ugrid_tune (A/B switches), repeatable")
    ap.add_argument("--sync-free", type=int, default=0, help="1: the native step without its mid-forward host read (ugrid_voxgo_step.sync_free)")
    ap.add_argument("--channels-last", type=int, default=1, help="k0 stored [P][X][Y][Z][C] (the training layout) or row-major")
    return ap.parse_args(argv)


def run(args):
    """One measurement; returns the result dict (bench.py embeds it as `secondary_s3_train_step`)."""
    from unboundednerfpytorch_amd import train_step as ts
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model
    from unboundednerfpytorch_amd import _gradpool, _lib
    for kv in getattr(args, "tune", []):
        k, v = kv.split("=")
        _lib.check(_lib.load().ugrid_tune(k.encode(), int(v)), "tune " + kv)
    _gradpool.touch_enabled = bool(getattr(args, "touch", 1))
    dev = torch.device("cuda", 0)
    for _ in range(int(getattr(args, "burn_streams", 0))):
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            torch.zeros(1, device=dev)
    model = make_model(args.grid, args.freq, dev, args.fused, args.channels_last)
    model.fused_loss = bool(args.fused_loss)
    model.native_sync_free = bool(getattr(args, "sync_free", 0))
    opt = create_optimizer_or_freeze_model(model, TRUCK_CFG, global_step=0)
    rk = dict(stepsize=0.5, rand_bkgd=True)
    timers = None
    stats = {}
    import contextlib
    import time
    # with the k0 update on a second stream, the iteration itself runs on a HIGH-priority stream: its short latency-bound
    # kernels are scheduled ahead of the update's 844k bandwidth-bound workgroups instead of queueing behind them
    main = torch.cuda.Stream(priority=-1) if args.overlap else None
    if main is not None:
        main.wait_stream(torch.cuda.current_stream())
    batches = [random_rays(args.rays, dev, seed=step) for step in range(1, args.warmup + args.steps + 1)]   # ray sampling is not the step
    # --blocks B: the timed steps are clocked in B equal blocks (a synchronise + host clock between them) and the MEDIAN block is
    # reported -- on these boxes ~5 % of host-synchronised starts lose 25-80 ms to an idle -> busy transition of the queue
    # (profiles/r05/share_stall_diag.txt), which a 20-step mean carries as +1-4 ms per step
    n_blocks = max(1, int(getattr(args, "blocks", 1)))
    per_block = max(1, args.steps // n_blocks)
    block_ms = []
    for step in range(1, args.warmup + args.steps + 1):
        k = step - args.warmup - 1
        if k >= 0 and k % per_block == 0 and k // per_block < n_blocks:
            torch.cuda.synchronize()
            now = time.perf_counter()
            if k == 0:
                timers = {}
                t0 = now
            else:
                block_ms.append((now - tb) * 1e3 / per_block)
            tb = now
        o, d, v, rgb = batches[step - 1]
        with (torch.cuda.stream(main) if main is not None else contextlib.nullcontext()):
            loss, psnr = ts.train_iteration(model, opt, o, d, v, rgb, TRUCK_CFG, args.first_step - 1 + step, rk, timers=timers,
                                            overlap_k0_update=bool(args.overlap), return_tensors=bool(getattr(args, "lazy_loss", 0)))
        stats = {"loss": loss, "psnr": psnr}
    torch.cuda.synchronize()
    now = time.perf_counter()
    wall_ms = (now - t0) * 1e3 / args.steps
    if n_blocks > 1 and args.steps == per_block * n_blocks:
        block_ms.append((now - tb) * 1e3 / per_block)
        wall_ms = sorted(block_ms)[len(block_ms) // 2]
    stats = {k: float(v) for k, v in stats.items()}
    phases = ["forward", "loss", "backward", "tv+adam"]
    order = ["start"] + phases
    ms = {}
    for a, b in zip(order[:-1], order[1:]):
        ms[b] = sum(x.elapsed_time(y) for x, y in zip(timers[a], timers[b])) / args.steps
    total = wall_ms        # all streams, host clock around the timed steps; the phase split is the main stream's
    # fraction of the k0 gradient's 256-byte lines one backward marks (a fresh backward; the optimizer clears the bitmap)
    touched = None
    if _gradpool.touch_enabled:
        _gradpool.certify([model.k0.grid])           # (what train_iteration does for its own backward)
        out = model(o, d, v, global_step=step, is_train=True, **rk)
        out["rgb_marched"].sum().backward()
        tb = _gradpool.touch_of(model.k0.grid, model.k0.grid.grad)
        _gradpool.decertify([model.k0.grid])
        if tb is not None:
            w = tb.to(torch.int64) & 0xFFFFFFFF
            bits = sum(int(((w >> i) & 1).sum()) for i in range(32))
            touched = bits / float((model.k0.grid.numel() + 63) // 64)
    with torch.no_grad():
        out = model(o, d, v, global_step=step, is_train=True, **rk)
    hbm_roof = tv_adam_dense_roofline(model, opt, dev)
    M = int(out["weights"].numel())
    S = int(out["n_max"])
    n_k0 = model.k0.grid.numel()
    res = {"workload": "S3: truck_single-shaped train step, P=%d, G=%d^3, C=12, %d random rays x S=%d, stepsize 0.5, dense TV + masked Adam"
                       % (1 + 2 * args.freq, args.grid, args.rays, S),
           "fused_forward": bool(getattr(model, "fused_forward", False)),
           "k0_channels_last": not model.k0.grid.is_contiguous(), "fused_loss": bool(args.fused_loss), "overlap_k0_update": bool(args.overlap),
           "tv_phase": "dense" if args.first_step + args.warmup + args.steps - 1 < TRUCK_CFG["tv_dense_before"] else "masked",
           "touch_bitmap": bool(_gradpool.touch_enabled), "lazy_loss": bool(getattr(args, "lazy_loss", 0)), "sync_free": bool(getattr(args, "sync_free", 0)), "k0_grad_lines_touched_frac": touched,
           "side_stream_pick": getattr(__import__("unboundednerfpytorch_amd.sharded_adam", fromlist=["x"])._low_priority_stream, "last", None),
           "ms_per_step": total, "block_ms": [round(x, 4) for x in block_ms] if n_blocks > 1 else None, "phases_ms": ms, "steps": args.steps, "survivors_M": M, "samples": args.rays * S,
           "rays_per_sec": args.rays / (total * 1e-3), "k0_voxels": n_k0,
           "k0_streaming_floor_ms": {"note": "compulsory HBM passes over the 3.46 GB k0-sized arrays per step at 6.3 TB/s achievable: "
                                             "grad zero-fill (1x write), dense TV (param read + grad read/write), masked Adam (grad read)",
                                     "value": 5 * n_k0 * 4 / 6.3e12 * 1e3},
           "roofline_tv_adam_dense": hbm_roof,
           **stats}
    return res


def tv_adam_dense_roofline(model, opt, dev, reps=6):
    """The one genuinely HBM-bound kernel of the path: the fused dense TV + Adam pass over the k0 grid
    (ugrid_tv_adam_dense_cl; run_train.py:281-288 as one kernel).  Algorithmic bytes per launch = 7 arrays x numel x 4 B: it reads
    param, grad, exp_avg, exp_avg_sq and writes param_out, exp_avg, exp_avg_sq -- each byte once (the stencil's neighbours come
    from cache).  Timed alone with HIP events on the launch stream; peak = the guide's 8 TB/s HBM3E figure."""
    from unboundednerfpytorch_amd import _lib, adam_upd_cuda
    _lib.wait_pending(model.k0.grid)
    torch.cuda.synchronize()
    p = model.k0.grid.data
    st = opt.state.get(model.k0.grid, {})
    if "exp_avg" not in st or st["exp_avg"].shape != p.shape:
        return None
    alt = torch.empty_like(p, memory_format=torch.preserve_format)
    g = torch.zeros_like(p, memory_format=torch.preserve_format)
    from unboundednerfpytorch_amd.sharded_adam import ShardedMaskedAdam
    ShardedMaskedAdam._flat(g)[::4099] = 1e-3          # a sparse gradient (the masked rule still reads every element)
    m, v = st["exp_avg"].clone(memory_format=torch.preserve_format), st["exp_avg_sq"].clone(memory_format=torch.preserve_format)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ok = True
    for i in range(reps + 1):
        if i:
            ev[i - 1].record()
        ok = ok and adam_upd_cuda.tv_adam_dense(p if i % 2 == 0 else alt, alt if i % 2 == 0 else p, g, m, v, 1e-7, 1e-7, 1e-7, 5 + i, 0.9, 0.99,
                                                1e-9, 1e-8, True)
    ev[reps].record()
    torch.cuda.synchronize()
    if not ok:
        return None
    if (reps + 1) % 2 == 1:                       # an odd number of swaps: the current values sit in `alt`
        p.copy_(alt)
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]
    nbytes = 7 * p.numel() * 4
    del alt, g, m, v
    torch.cuda.empty_cache()
    return {"kernel": "ugrid_tv_adam_dense_cl" if not p.is_contiguous() else "ugrid_tv_adam_dense", "bound": "hbm", "ms": ms,
            "algorithmic_bytes": nbytes, "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
            "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0,
            "note": "7 k0-sized arrays x 4 B, each byte once; median of %d launches, HIP events on the launch stream" % reps}


def main():
    print(json.dumps(run(parse())))


if __name__ == "__main__":
    main()
