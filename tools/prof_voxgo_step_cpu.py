"""Host-side profile (cProfile) of the DirectVoxGO / DirectContractedVoxGO train step: where the Python / launch time goes.
    python tools/prof_voxgo_step_cpu.py [dvgo|dcvgo]        (GPU box)"""
import argparse
import cProfile
import io
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import bench_train_step as bts  # noqa: E402
import bench_voxgo_train as bvt  # noqa: E402
from unboundednerfpytorch_amd import train_step as ts  # noqa: E402
from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "dvgo"
    dev = torch.device("cuda", 0)
    cfg = bvt.CFG[kind]
    model = bvt.make_model(kind, 160 if kind == "dvgo" else 320, dev, 1)
    opt = create_optimizer_or_freeze_model(model, cfg, global_step=0)
    rk = dict(stepsize=0.5, bg=1, near=0.2, far=6.0) if kind == "dvgo" else dict(stepsize=0.5, bg=1, rand_bkgd=True)
    first = 1 if kind == "dvgo" else 10001
    rays = [bts.random_rays(cfg["N_rand"], dev, seed=s) for s in range(1, 6)]
    for s in range(4):
        ts.train_iteration(model, opt, *rays[s % 5], cfg, first + s, rk, overlap_k0_update=True)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for s in range(4, 4 + n):
        ts.train_iteration(model, opt, *rays[s % 5], cfg, first + s, rk, overlap_k0_update=True)
    torch.cuda.synchronize()
    print("unprofiled: %.3f ms per step" % ((time.perf_counter() - t0) * 1e3 / n))
    pr = cProfile.Profile()
    pr.enable()
    for s in range(4 + n, 4 + 2 * n):
        ts.train_iteration(model, opt, *rays[s % 5], cfg, first + s, rk, overlap_k0_update=True)
    torch.cuda.synchronize()
    pr.disable()
    for key in ("cumulative", "tottime"):
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(40)
        txt = buf.getvalue().replace("/root/repo/", "")
        print("\n".join(l[:170] for l in txt.splitlines()[4:]))
    print("steps", n)


if __name__ == "__main__":
    main()
