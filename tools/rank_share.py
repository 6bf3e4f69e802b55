"""Render rank r's share of an N-way deal of the S1 frame ALONE on this GPU, a few times (for a rocprofv3 --pmc pass:
tools/gpu_rank_share.sh).  Same FrameBench path as bench.py's `scaling_proxy`."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--ranks", default="0")
    ap.add_argument("--deal", default="tiles", choices=["tiles", "rows", "bands"])
    ap.add_argument("--frames", type=int, default=3)
    a = ap.parse_args()
    import bench
    dev = torch.device("cuda", 0)
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    args.contiguous, args.deal, args.deal_group = False, a.deal, 0
    state = bench.make_state(args.grid, dev, seed=0)
    fb0 = bench.FrameBench(args, state, dev, 1, 0, None)
    del state
    torch.cuda.empty_cache()
    out = {}
    for r in [int(x) for x in a.ranks.split(",")]:
        fb = bench.FrameBench(args, None, dev, a.n, r, None, renderer=fb0.rend) if a.n > 1 else fb0
        dt, timing = fb.timed(a.frames, 1)
        out[r] = {"ms": dt / a.frames * 1e3, "kernels_ms": bench.kernel_ms(timing, a.frames), "rays": sum(n for _, n in timing) // a.frames}
    print(json.dumps({"n": a.n, "deal": a.deal, "shares": out}))


if __name__ == "__main__":
    main()
