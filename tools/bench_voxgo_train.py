"""Training-step time of the two dense-grid models at their configurations' sizes, through
voxgo_model + train_step.train_iteration (the loop body of run_train.py:185-296):

  dvgo   nerf_synthetic 'lego' fine stage (configs/default.py fine_*: 160^3, rgbnet_dim 12, N_rand 8192, stepsize 0.5,
         entropy_last 1e-3, rgbper 1e-2, no TV) -- the model of BASELINE.json configs[0]
  dcvgo  Mip-NeRF-360 fine stage (configs/nerf_unbounded/nerf_unbounded_default.py: 320^3 contracted grid, bg_len 0.2, N_rand 4096,
         distortion 1e-2, entropy_last 1e-3, rgbper 1e-2, TV density 1e-6 / k0 1e-7, dense before 10 000) -- configs[1]'s model

    python tools/bench_voxgo_train.py [--model dvgo|dcvgo|both] [--steps 20] [--fused 0|1]          (GPU box)

One JSON line per (model, TV phase): ms per step (host clock around the timed steps, all streams), survivors, rays / s.
Synthetic trained-like fields (bench.make_state_surfaces: smooth occupancy with opaque surfaces), random rays; `--fused 0`
runs the op-by-op chain over the same drop-in ops for comparison."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

CFG = {
    "dvgo": dict(N_rand=8192, weight_main=1.0, weight_entropy_last=1e-3, weight_rgbper=1e-2, weight_nearclip=0.0, weight_distortion=0.0,
                 weight_tv_density=0.0, weight_tv_k0=0.0, tv_before=0, tv_dense_before=0, tv_after=0, tv_every=1, lrate_density=1e-1,
                 lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_decay=20, skip_zero_grad_fields=['density', 'k0'], pg_scale=[]),
    "dcvgo": dict(N_rand=4096, weight_main=1.0, weight_entropy_last=1e-3, weight_rgbper=1e-2, weight_nearclip=0.0, weight_distortion=1e-2,
                  weight_tv_density=1e-6, weight_tv_k0=1e-7, tv_before=1e9, tv_dense_before=10000, tv_after=0, tv_every=1,
                  lrate_density=1e-1, lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_decay=80, skip_zero_grad_fields=['density', 'k0'], pg_scale=[]),
}


def make_model(kind, G, dev, fused):
    import bench
    from unboundednerfpytorch_amd import voxgo_model as vm
    if kind == "dvgo":
        m = vm.DirectVoxGO(xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1], num_voxels=G ** 3, num_voxels_base=G ** 3, alpha_init=1e-2,
                           fast_color_thres=1e-4, rgbnet_dim=12, rgbnet_direct=True).to(dev)
    else:
        m = vm.DirectContractedVoxGO(xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1], num_voxels=G ** 3, num_voxels_base=G ** 3, alpha_init=1e-4,
                                     fast_color_thres=1e-4, rgbnet_dim=12).to(dev)
    m.fused_forward = bool(fused)
    m.fused_rgbnet = bool(fused)
    m.fused_loss = bool(fused)
    Gd = int(m.world_size[0])
    st = bench.make_state_surfaces(Gd, dev, seed=0)
    with torch.no_grad():
        g = torch.Generator(device=dev)
        g.manual_seed(5)
        m.density.grid.copy_(st["density_grid"][:1] / 7.0)     # (level 0 of the F = 3 recipe carries 7 x the occupancy field)
        m.k0.grid.normal_(0.0, 0.5, generator=g)
        m.k0.grid += st["k0_grid"][:1]
        m.update_occupancy_cache()
    del st
    torch.cuda.empty_cache()
    return m


def run(kind, args, first_step):
    from bench_train_step import random_rays
    from unboundednerfpytorch_amd import train_step as ts
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model
    dev = torch.device("cuda", 0)
    cfg = CFG[kind]
    G = args.grid or (160 if kind == "dvgo" else 320)
    m = make_model(kind, G, dev, args.fused)
    m.native_step = bool(getattr(args, "native", 1))
    m.native_sync_free = bool(getattr(args, "sync_free", 0))
    opt = create_optimizer_or_freeze_model(m, cfg, global_step=0)
    rk = dict(stepsize=0.5, bg=1, near=0.2, far=6.0) if kind == "dvgo" else dict(stepsize=0.5, bg=1, rand_bkgd=True)
    n = cfg["N_rand"]
    batches = [random_rays(n, dev, seed=s) for s in range(1, args.warmup + args.steps + 1)]
    # --blocks B: B equal blocks, the median block is reported (tools/bench_train_step.py: the queue's idle -> busy glitch)
    n_blocks = max(1, int(getattr(args, "blocks", 1)))
    per_block = max(1, args.steps // n_blocks)
    block_ms = []
    for step in range(1, args.warmup + args.steps + 1):
        k = step - args.warmup - 1
        if k >= 0 and k % per_block == 0 and k // per_block < n_blocks:
            torch.cuda.synchronize()
            now = time.perf_counter()
            if k == 0:
                t0 = now
            else:
                block_ms.append((now - tb) * 1e3 / per_block)
            tb = now
        o, d, v, rgb = batches[step - 1]
        loss, psnr = ts.train_iteration(m, opt, o, d, v, rgb, cfg, first_step - 1 + step, rk, overlap_k0_update=bool(args.overlap),
                                        return_tensors=bool(args.lazy_loss))
    torch.cuda.synchronize()
    now = time.perf_counter()
    ms = (now - t0) * 1e3 / args.steps
    if n_blocks > 1 and args.steps == per_block * n_blocks:
        block_ms.append((now - tb) * 1e3 / per_block)
        ms = sorted(block_ms)[len(block_ms) // 2]
    with torch.no_grad():
        out = m(o, d, v, global_step=step, is_train=True, **rk)
    tv_on = cfg["weight_tv_k0"] > 0
    return {"model": kind, "workload": "%s train step: G=%s, C=12, %d random rays, stepsize 0.5%s" % (
                "DirectVoxGO (lego fine-stage shape)" if kind == "dvgo" else "DirectContractedVoxGO (mip-360 fine-stage shape)",
                m.world_size.tolist(), n, "" if not tv_on else ", TV " + ("dense" if first_step < cfg["tv_dense_before"] else "masked")),
            "fused": bool(args.fused), "native_step": bool(m.native_step and args.fused), "sync_free": bool(m.native_sync_free), "lazy_loss": bool(args.lazy_loss), "ms_per_step": ms, "block_ms": [round(x, 4) for x in block_ms] if n_blocks > 1 else None, "rays_per_sec": n / (ms * 1e-3), "survivors_M": int(out["weights"].numel()),
            "mask_cache_occupied_frac": float(m.mask_cache.mask.float().mean()), "steps": args.steps, "loss": float(loss), "psnr": float(psnr)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="both")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--overlap", type=int, default=1)
    ap.add_argument("--native", type=int, default=1, help="0: the op-by-op fused step (four autograd nodes issued from Python) instead of "
                    "native_step.VoxGOStep (one node, three C calls) -- the same kernels and bits")
    ap.add_argument("--phase", default="both", help="dcvgo: dense | masked | both TV phases")
    ap.add_argument("--lazy-loss", type=int, default=0, help="train_iteration(return_tensors=True): no host read of loss / psnr per step "
                    "(the reference reads psnr.item() every step; a caller that logs every N steps need not)")
    ap.add_argument("--sync-free", type=int, default=0, help="1: the native step without its mid-forward host read (capacity-sized per-sample arrays, counts on "
                    "the device: include/ugrid_hip.h ugrid_voxgo_step.sync_free); with --lazy-loss 1 the loop makes no host read at all")
    ap.add_argument("--blocks", type=int, default=1, help="clock the timed steps in B equal blocks and report the median block")
    ap.add_argument("--tune", action="append", default=[], help="key=value for ugrid_tune (A/B switches), repeatable")
    args = ap.parse_args()
    from unboundednerfpytorch_amd import _lib
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(_lib.load().ugrid_tune(k.encode(), int(v)), "tune " + kv)
    kinds = ["dvgo", "dcvgo"] if args.model == "both" else [args.model]
    for kind in kinds:
        phases = [1] if CFG[kind]["weight_tv_k0"] == 0 else {"dense": [1], "masked": [10001]}.get(args.phase, [1, 10001])
        for first in phases:
            print(json.dumps(run(kind, args, first)), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
