"""Is the sync-free training step bound by the GPU or by the host?  Forward + fused loss + backward of the native step (no optimizer) at
the configurations' sizes (tools/bench_voxgo_train.py's models), issued eagerly and replayed from a hipGraph of the same step.

    python tools/bench_graph_step.py [--model dvgo|dcvgo|both] [--steps 200]          (GPU box)

One JSON line per model: ms per step eager host-counted / eager sync-free / graph replay."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def clock(fn, steps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


def run(kind, steps):
    from bench_train_step import random_rays
    from bench_voxgo_train import CFG, make_model
    from unboundednerfpytorch_amd.ops import loss_coefficients
    dev = torch.device("cuda", 0)
    cfg = CFG[kind]
    m = make_model(kind, 160 if kind == "dvgo" else 320, dev, 1)
    m.native_step = True
    rk = dict(stepsize=0.5, bg=1, near=0.2, far=6.0) if kind == "dvgo" else dict(stepsize=0.5, bg=1)
    R = cfg["N_rand"]
    o, d, v, rgb = random_rays(R, dev, seed=1)
    coef = loss_coefficients(cfg, R, m.sample_table(rk["stepsize"], dev).numel(), None, 1)
    kw = dict(rk, fused_loss={'target': rgb, 'coef': coef})
    side = torch.cuda.Stream()

    def step():
        m.zero_grad(set_to_none=True)
        out = m(o, d, v, global_step=1, is_train=True, **kw)
        out["loss"].backward()
        return out

    res = {"model": kind, "rays": R}
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):            # (eager work before a capture stays off the default stream: profiles/r06/NOTES.md M-O)
        m.native_sync_free = False
        res["eager_host_counted_ms"] = clock(step, steps)
        m.native_sync_free = True
        res["eager_sync_free_ms"] = clock(step, steps)
        out = step()
        res["samples"] = out["native"]["out"]["n_valid"].tolist()
        del out
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    m.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(o, d, v, global_step=1, is_train=True, **kw)
        out["loss"].backward()
    res["graph_replay_ms"] = clock(g.replay, steps)
    res["loss"] = float(out["loss"])
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="both")
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    for kind in (("dvgo", "dcvgo") if a.model == "both" else (a.model,)):
        print(json.dumps(run(kind, a.steps)), flush=True)
