"""Why do some rank shares of bench.scaling_proxy take 20 ms of host time while their kernels take 1.4?  (one visit; not a
product tool)  Per step of a dealt share: GPU time between two events around the render, host time until an event-query poll
sees the end, and the time torch.cuda.synchronize() / stream.synchronize() take -- GPU-side hole or host-side wake-up?"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "poll"
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda", 0)
state = bench.make_state(args.grid, dev, seed=0)
fb0 = bench.FrameBench(args, state, dev, 1, 0, None)
del state
torch.cuda.empty_cache()
fb0.timed(3, 1)
res = []
for N, deal_group, contiguous in ((8, 1, False), (8, 1, True)):
    a = argparse.Namespace(**dict(vars(args), contiguous=contiguous, deal="tiles", deal_group=deal_group))
    for r in range(N):
        fb = bench.FrameBench(a, None, dev, N, r, None, renderer=fb0.rend)
        ro, rd, vd = fb.get_rays_idx(fb.H, fb.W, fb.K, fb.c2w, fb.px)
        torch.cuda.synchronize()
        rows = []
        for i in range(24):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            o = fb.rend(ro, rd, vd, stepsize=fb.stepsize, render_depth=True, **fb.rkw)
            e1.record()
            t1 = time.perf_counter()
            if mode == "poll":
                while not e1.query():
                    pass
            elif mode == "stream":
                torch.cuda.current_stream().synchronize()
            else:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            rows.append((round(e0.elapsed_time(e1), 3), round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3), round((t3 - t2) * 1e3, 3)))
        slow = [x for x in rows if x[2] > 5 or x[0] > 5]
        print(mode, "N=%d group=%d contiguous=%s rank=%d" % (N, deal_group, contiguous, r), "slow (gpu_ms, issue, wait, resync):", slow, "typical", rows[-1], flush=True)
        res.append({"mode": mode, "N": N, "contiguous": contiguous, "rank": r, "rows": rows})
        del fb
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "diag_share_stall_%s.json" % mode), "w"))
