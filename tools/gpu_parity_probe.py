"""Where does the largest GPU-vs-oracle difference on the S1 frame come from?  Renders one 8192-ray chunk of the
bench frame on the GPU and with the CPU oracle, then prints the worst rays per output with their threshold margin
and how many survivors they have.  python tools/gpu_parity_probe.py [start_ray]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import model_oracle
from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_a_view

dev = torch.device("cuda:0")
G, H, W = 200, 1080, 1920
state = bench.make_state(G, dev, seed=0)
K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in get_rays_of_a_view(H, W, K, bench.camera(0, dev))]
b = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = 8192
mode = int(os.environ.get("MLP_MODE", "-1"))
rend = FourierGridRenderer(state, dev, mlp_mode=None if mode < 0 else mode)
out = rend(ro[b:b + n], rd[b:b + n], vd[b:b + n], stepsize=1.31, render_depth=True)
cpu = {k: ([x.cpu() for x in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in state.items()}
torch.set_num_threads(8)
ref = model_oracle.fouriergrid_render(cpu, ro[b:b + n].cpu(), rd[b:b + n].cpu(), vd[b:b + n].cpu(), 1.31, render_depth=True, return_margin=True)
nsurv = torch.bincount(ref["ray_id"], minlength=n)
for k in ("rgb_marched", "depth", "alphainv_last"):
    err = (out[k].cpu() - ref[k]).abs()
    err = err.amax(dim=1) if err.dim() == 2 else err
    top = torch.topk(err, 5)
    print(k, "mean %.2e  p99 %.2e  max %.2e" % (float(err.mean()), float(err.kthvalue(int(0.99 * n)).values), float(err.max())))
    for e, i in zip(top.values.tolist(), top.indices.tolist()):
        print("   ray %5d err %.3e margin %.2e survivors %3d  ref %s gpu %s  alphainv_last ref %.6f" % (
            i, e, float(ref["margin"][i]), int(nsurv[i]), ref[k][i].tolist(), out[k][i].cpu().tolist(), float(ref["alphainv_last"][i])))
# survivors of the worst rgb ray: weights around it
i = int(torch.topk((out["rgb_marched"].cpu() - ref["rgb_marched"]).abs().amax(dim=1), 1).indices)
sel = ref["ray_id"] == i
print("worst rgb ray", i, "weights", ref["weights"][sel].tolist()[:40])
print("raw_alpha", ref["raw_alpha"][sel].tolist()[:40])
