"""S5 (BASELINE.json configs[4]): Waymo Block-NeRF-style rendering -- one FourierGrid BLOCK model per GPU
at the real block shape of configs/waymo/waymo_no_block.py:129-149 (G = 300^3, rgbnet_dim = 3, viewbase_pe = 2,
contracted_norm = 'l2', stepsize 0.5 -> S = 1002), every rank renders ITS block for ALL rays of the frame and
dist.composite_blocks merges them with one all-reduce (the repository's only block-merging rule, legacy
eval_block_nerf.py:95-133).

    python tools/bench_s5_blocks.py [--steps 5]                                   (1 GPU = 1 block)
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_s5_blocks.py     (N blocks)

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
    args = ap.parse_args()
    import bench
    from unboundednerfpytorch_amd.dist import composite_blocks
    from unboundednerfpytorch_amd.fourier_render import (FourierGridRenderer, get_rays_of_a_view, get_rays_of_pixel_index,
                                                         pixel_tile_order, untile)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
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
    ang = 2 * math.pi * rank / max(world, 1)
    centroid = [0.8 * math.cos(ang), 0.8 * math.sin(ang), 0.0]                        # block centroids on a ring

    def frame():
        out = composite_blocks(rend.forward, ro, rd, vd, cam, centroid, stepsize=0.5)
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
    if rank == 0:
        print(json.dumps({
            "workload": "S5: %d block model(s), one per GPU: G=%d^3, F=3, C=3, viewbase_pe=2, l2 contraction, %dx%d rays x S=%d, "
                        "composited with one all-reduce of [R,7] (dist.composite_blocks)" % (world, G, W, H, S),
            "n_gpus": world, "ms_per_frame": dt * 1e3, "value": world * R * S / dt / 1e6, "unit": "Msamples/s (summed over blocks)",
            "rays_per_sec": R / dt, "survivors_block0": M, "block_weight_rank0": out["block_weight"],
            "finite": bool(torch.isfinite(out["rgb_marched"]).all())}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
