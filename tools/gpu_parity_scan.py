"""Scan the 12 sampled chunks of the S1 bench frame for the rays where the GPU and the fp32 CPU oracle disagree most,
then evaluate those rays in fp64 (tools/parity_fp64.py): is the difference the GPU's error, or the sensitivity of the
ray itself to fp32 rounding (oracle32 vs fp64 just as far apart)?  Output -> profiles/r01/parity_scan_s1.txt"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from oracle import model_oracle
from parity_fp64 import render_fp64
from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_a_view

dev = torch.device("cuda:0")
G, H, W, n_chunks, chunk = 200, 1080, 1920, 12, 8192
state = bench.make_state(G, dev, seed=0)
K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in get_rays_of_a_view(H, W, K, bench.camera(0, dev))]
R = ro.shape[0]
rend = FourierGridRenderer(state, dev)
out = rend(ro, rd, vd, stepsize=1.31, render_depth=True)
out = {k: out[k].cpu() for k in ("rgb_marched", "depth", "alphainv_last")}
cpu = {k: ([x.cpu() for x in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in state.items()}
del state, rend
torch.cuda.empty_cache()
torch.set_num_threads(8)
ro, rd, vd = ro.cpu(), rd.cpu(), vd.cpu()
starts = [int(i * (R - chunk) / (n_chunks - 1)) // 64 * 64 for i in range(n_chunks)]
cands = []
for b in starts:
    ref = model_oracle.fouriergrid_render(cpu, ro[b:b + chunk], rd[b:b + chunk], vd[b:b + chunk], 1.31, render_depth=True, return_margin=True)
    err = torch.zeros(chunk)
    for k in out:
        e = (out[k][b:b + chunk] - ref[k]).abs()
        err = torch.maximum(err, e.amax(dim=1) if e.dim() == 2 else e)
    top = torch.topk(err, 2)
    for e, i in zip(top.values.tolist(), top.indices.tolist()):
        cands.append((e, b + i, float(ref["margin"][i]), {k: ref[k][i].clone() for k in out}))
cands.sort(key=lambda c: -c[0])
grids64 = (cpu["density_grid"].double(), cpu["k0_grid"].double())
print("worst rays of %d sampled (GPU vs fp32 oracle), re-evaluated in fp64:" % (n_chunks * chunk))
for e, i, margin, ref in cands[:5]:
    f64 = render_fp64(cpu, ro[i:i + 1], rd[i:i + 1], vd[i:i + 1], 1.31, grids64)
    eg = max(float((out[k][i].double() - f64[k][0]).abs().max()) for k in out)
    eo = max(float((ref[k].double() - f64[k][0]).abs().max()) for k in out)
    print("ray %7d  margin %.2e  |gpu-oracle32| %.2e   |gpu-fp64| %.2e   |oracle32-fp64| %.2e   alphainv_last gpu %.6f oracle32 %.6f fp64 %.6f"
          % (i, margin, e, eg, eo, float(out["alphainv_last"][i]), float(ref["alphainv_last"]), float(f64["alphainv_last"][0])))
