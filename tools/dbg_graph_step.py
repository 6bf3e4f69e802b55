"""Bisect of the hipGraph capture of the sync-free training step (round 6, visit H): each stage in its own process (a crash in
hipStreamEndCapture must not take the other stages with it).   python tools/dbg_graph_step.py [stage kind]"""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stage(name, kind):
    import torch
    import test_gpu_voxgo_train as T
    faulthandler.enable()
    dev = torch.device("cuda", 0)
    m, (o, d, v), kw, R = T._sync_free_pair(kind, dev)
    if "seedonly" in name:
        torch.manual_seed(5)
    if "ref" in name:                 # what the test does first: eager reference runs on the same model, default stream
        for _ in range(2):
            m.native_sync_free = ("sfref" in name)            # sfref: the eager runs are sync-free too (is it the host count, or any eager run?)
            m.zero_grad(set_to_none=True)
            if "noseed" not in name:
                torch.manual_seed(5)
            r = m(o, d, v, global_step=1, is_train=True, **kw)
            if "fwdonlyref" not in name:
                r["loss"].backward()
        m.zero_grad(set_to_none=True)
        if "gc" in name:
            import gc
            del r
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        print(name, kind, "eager references done", flush=True)
    if "clone" in name:               # static clones as the graph's inputs
        o, d, v = o.clone(), d.clone(), v.clone()
        kw = dict(kw, fused_loss=dict(kw["fused_loss"], target=kw["fused_loss"]["target"].clone()))
    m.native_sync_free = {'hints': (0, 0)}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            w = m(o, d, v, global_step=1, is_train=True, **kw)
            w["loss"].backward()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    m.zero_grad(set_to_none=True)
    print(name, kind, "warm-up done", flush=True)
    g = torch.cuda.CUDAGraph()
    mode = "thread_local" if "tl" in name else "global"
    with torch.cuda.graph(g, capture_error_mode=mode):
        out = m(o, d, v, global_step=1, is_train=True, **kw)
        print(name, "forward captured", flush=True)
        if "bwd" in name:
            out["loss"].backward()
            print(name, "backward captured", flush=True)
            grads = {k: p.grad for k, p in m.named_parameters()}
    print(name, "capture ended", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(name, kind, "replayed: loss", float(out["loss"]), "n_valid", out["native"]["out"]["n_valid"].tolist(), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        stage(sys.argv[1], sys.argv[2])
    else:
        for kind in ("dvgo",):
            for name in ("fwdonlyref_noseed_fwd_bwd", "sfref_noseed_fwd_bwd", "gc_ref_noseed_fwd_bwd"):
                r = subprocess.run([sys.executable, __file__, name, kind], capture_output=True, text=True, timeout=300)
                tail = [l for l in (r.stdout + r.stderr).splitlines() if "Warning" not in l and "amdgpu.ids" not in l]
                print("==", kind, name, "rc", r.returncode)
                print("\n".join(tail[:14]))
