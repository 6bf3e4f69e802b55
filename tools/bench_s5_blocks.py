"""S5 (BASELINE.json configs[4]): Waymo Block-NeRF-style rendering -- one FourierGrid BLOCK model per GPU
at the real block shape of configs/waymo/waymo_no_block.py:129-149 (G = 300^3, rgbnet_dim = 3, viewbase_pe = 2,
contracted_norm = 'l2', stepsize 0.5 -> S = 1002), every rank renders ITS block for ALL rays of the frame and
dist.composite_blocks merges them with one all-reduce (the repository's only block-merging rule, legacy
eval_block_nerf.py:95-133).

    python tools/bench_s5_blocks.py [--steps 5]                                   (1 GPU = 1 block)
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_s5_blocks.py     (N blocks)
    UGRID_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 ... tools/bench_s5_blocks.py --check
        two REAL blocks (different seeds, different centroids) as two gloo ranks sharing the one GPU of the box
        (timings then only say that it runs); --check: rank 0 also renders BOTH blocks on its own and evaluates the
        merging rule (eval_block_nerf.py:95-133: drop blocks whose mean visibility <= 0.05, normalised inverse
        distance^4 weights) in a single process; the composited frame must agree.

Prints one JSON line on rank 0: ms per composited frame, Msamples/s summed over the blocks."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=300)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--check", action="store_true", help="rank 0 re-evaluates the merging rule over all blocks in one process and compares")
    ap.add_argument("--out", default=None, help="also write the JSON line to this file")
    args = ap.parse_args()
    import bench
    from unboundednerfpytorch_amd.dist import composite_blocks
    from unboundednerfpytorch_amd.fourier_render import (FourierGridRenderer, get_rays_of_a_view, get_rays_of_pixel_index,
                                                         pixel_tile_order, untile)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("UGRID_BENCH_SHARE_GPU") == "1"      # debugging: all ranks on the devices present, over gloo
    if share:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    G, H, W = args.grid, args.height, args.width
    state = bench.make_state_surfaces(G, dev, seed=rank, C=3, pe=2, norm="l2")       # one block per rank, its own seed
    rend = FourierGridRenderer(state, dev)
    del state
    torch.cuda.empty_cache()
    K = [[1600.0 * W / 1920.0, 0, W / 2.0], [0, 1600.0 * W / 1920.0, H / 2.0], [0, 0, 1]]
    c2w = bench.camera(0, dev)
    order = pixel_tile_order(H, W, dev)          # 8 x 8 pixel blocks, one per march wave (what render_view does)
    if order is not None:
        ro, rd, vd = get_rays_of_pixel_index(H, W, K, c2w, order)
    else:
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in get_rays_of_a_view(H, W, K, c2w)]
    R = ro.shape[0]
    S = rend.tables(0.5)[2]
    cam = c2w[:, 3].tolist()
    def centroid_of(b):                                                               # block centroids on a ring
        ang = 2 * math.pi * b / max(world, 1) + 0.4
        return [0.8 * math.cos(ang), 0.8 * math.sin(ang), 0.0]
    centroid = centroid_of(rank)

    def frame():
        out = composite_blocks(rend.forward, ro, rd, vd, cam, centroid, stepsize=0.5,
                               ray_order="coherent" if order is not None else "auto")
        if order is not None:                    # the composited frame back in image order
            out = dict(out, **{k: untile(out[k], H, W) for k in ("rgb_marched", "depth", "alphainv_last") if k in out})
        return out

    for _ in range(args.warmup):
        out = frame()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = frame()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    M = rend.survivors_of_last_chunk()
    check = None
    if args.check and rank == 0:
        # single-process evaluation of the rule over ALL blocks (each block's renderer rebuilt here from its seed)
        keys = ("rgb_marched", "depth", "alphainv_last")
        outs, dws, vis = [], [], []
        for b in range(world):
            if b == rank:
                rb = rend
            else:
                stb = bench.make_state_surfaces(G, dev, seed=b, C=3, pe=2, norm="l2")
                rb = FourierGridRenderer(stb, dev)
                del stb
            ob = rb(ro, rd, vd, stepsize=0.5, render_depth=True)
            outs.append({k: (untile(ob[k], H, W) if order is not None else ob[k]).double().cpu() for k in keys})
            dws.append(float(torch.tensor([cam[i] - centroid_of(b)[i] for i in range(3)], dtype=torch.float64).norm() ** -4.0))
            vis.append(float((1.0 - ob["alphainv_last"]).mean()) > 0.05)
            if b != rank:
                del rb
                torch.cuda.empty_cache()
        wts = [dws[b] if (vis[b] or not any(vis)) else 0.0 for b in range(world)]
        tot = sum(wts)
        check = {"visible": vis, "weights_normalised": [w_ / tot for w_ in wts]}
        for k in keys:
            want = sum(outs[b][k] * (wts[b] / tot) for b in range(world))
            check["linf_" + k] = float((out[k].double().cpu() - want).abs().max())
            check["blocks_differ_linf_" + k] = float((outs[0][k] - outs[-1][k]).abs().max())
        check["ok"] = bool(max(check["linf_" + k] for k in keys) <= 2e-6 and int(out["visible_blocks"]) == sum(vis))
    if rank == 0:
        line = json.dumps({
            "workload": "S5: %d block model(s), one per GPU: G=%d^3, F=3, C=3, viewbase_pe=2, l2 contraction, %dx%d rays x S=%d, "
                        "composited with ONE all-reduce of [R+1,5] (dist.composite_blocks)%s" % (
                            world, G, W, H, S, "; DEBUG run: all ranks share one GPU over gloo, timings meaningless" if share and world > 1 else ""),
            "n_gpus": world, "ms_per_frame": dt * 1e3, "value": world * R * S / dt / 1e6, "unit": "Msamples/s (summed over blocks)",
            "rays_per_sec": R / dt, "survivors_block0": M, "block_weight_rank0": float(out["block_weight"]),
            "visible_blocks": int(out["visible_blocks"]), "check_vs_single_process_rule": check,
            "finite": bool(torch.isfinite(out["rgb_marched"]).all())})
        print(line)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            open(args.out, "w").write(line + "\n")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
