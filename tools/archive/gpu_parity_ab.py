"""One variant of the S1 parity study (driven by tools/gpu_parity_ab.sh with UGRID_LIB set): render the S1 frame,
compare 16 chunks x 8192 rays with the CPU oracle (evaluated once, cached in /tmp for the other variants), print
the per-output L-inf, the number of rays above 1e-4 and the worst rays' indices."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "?"
    import bench
    from oracle import model_oracle
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_a_view
    dev = torch.device("cuda", 0)
    G, H, W, n_chunks, chunk = 200, 1080, 1920, 16, 8192
    state = bench.make_state(G, dev, seed=0)
    rend = FourierGridRenderer(state, dev)
    K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
    ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in get_rays_of_a_view(H, W, K, bench.camera(0, dev))]
    R = ro.shape[0]
    out = rend(ro, rd, vd, stepsize=1.31, render_depth=True)
    torch.cuda.synchronize()
    starts = [int(i * (R - chunk) / (n_chunks - 1)) // 64 * 64 for i in range(n_chunks)]
    cache = "/tmp/s1_oracle_%d.pt" % n_chunks
    if os.path.exists(cache):
        ref = torch.load(cache)
    else:
        torch.set_num_threads(8)
        cpu_state = {k: ([x.cpu() for x in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in state.items()}
        parts = [model_oracle.fouriergrid_render(cpu_state, ro[b:b + chunk].cpu(), rd[b:b + chunk].cpu(), vd[b:b + chunk].cpu(), 1.31,
                                                 render_depth=True, return_margin=True) for b in starts]
        ref = {k: torch.cat([p[k] for p in parts]) for k in ("rgb_marched", "depth", "alphainv_last", "margin")}
        torch.save(ref, cache)
    idx = torch.cat([torch.arange(b, b + chunk) for b in starts])
    line = "%-11s" % name
    worst = []
    for k in ("rgb_marched", "depth", "alphainv_last"):
        err = (out[k].cpu()[idx] - ref[k]).abs()
        err = err.amax(dim=1) if err.dim() == 2 else err
        top = torch.topk(err, 3)
        worst.append((k, [(int(idx[i]), float(e), float(ref["margin"][i])) for e, i in zip(top.values, top.indices)]))
        line += "  %s linf %.3e >1e-4: %d" % (k.split("_")[0], float(err.max()), int((err > 1e-4).sum()))
    print(line + "   (%d rays)" % idx.numel())
    for k, w in worst:
        print("      %-14s worst rays (index, err, margin): %s" % (k, ", ".join("(%d, %.2e, %.1e)" % t for t in w)))


if __name__ == "__main__":
    main()
