"""The path's one HBM-bound kernel alone: ugrid_tv_adam_dense_cl (dense TV + masked Adam in one pass) over a k0 grid of S3's shape
(P = 9, C = 12, 200^3, channel-last = 3.46 GB per array).  Algorithmic bytes = 7 arrays x numel x 4 B (param, grad, exp_avg,
exp_avg_sq read; param_out, exp_avg, exp_avg_sq written; the stencil's neighbours come from cache).  Prints GB/s against the
guide's 8 TB/s; run under `rocprofv3 --pmc TCC_EA0_RDREQ_* / TCC_EA0_WRREQ_*` for the HBM bytes the launch really moved."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--levels", type=int, default=9)
    ap.add_argument("--channels", type=int, default=12)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--tune", action="append", default=[])
    a = ap.parse_args()
    from unboundednerfpytorch_amd import adam_upd_cuda
    from unboundednerfpytorch_amd.fourier_render import tune
    for kv in a.tune:
        tune(kv.split("=")[0], int(kv.split("=")[1]))
    dev = torch.device("cuda", 0)
    G = a.grid
    shape = (a.levels, a.channels, G, G, G)
    mk = lambda fill=None: (torch.empty(shape, device=dev).contiguous(memory_format=torch.channels_last_3d) if fill is None
                            else torch.full(shape, fill, device=dev).contiguous(memory_format=torch.channels_last_3d))
    p = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last_3d)
    alt, m, v = mk(), mk(0.0), mk(0.0)
    g = mk(0.0)
    g.permute(0, 2, 3, 4, 1).reshape(-1)[::4099] = 1e-3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
    for i in range(a.reps + 1):
        if i:
            ev[i - 1].record()
        ok = adam_upd_cuda.tv_adam_dense(p if i % 2 == 0 else alt, alt if i % 2 == 0 else p, g, m, v, 1e-7, 1e-7, 1e-7, 5 + i, 0.9, 0.99, 1e-9, 1e-8, True)
        assert ok
    ev[a.reps].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.reps))
    ms = ts[len(ts) // 2]
    nbytes = 7 * p.numel() * 4
    print(json.dumps({"kernel": "ugrid_tv_adam_dense_cl", "shape": list(shape), "ms_median": ms, "ms_min": ts[0], "algorithmic_bytes": nbytes,
                      "GBps": nbytes / (ms * 1e-3) / 1e9, "frac_of_8TBps": nbytes / (ms * 1e-3) / 1e9 / 8000.0, "tune": a.tune}))


if __name__ == "__main__":
    main()
