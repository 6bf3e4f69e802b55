"""Which torch thread count is fastest for the CPU-baseline oracle on this host? (GPU box has 256 cores)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from oracle import model_oracle
from unboundednerfpytorch_amd.fourier_render import get_rays_of_a_view
G = 200
state = bench.make_state(G, "cpu", 0)
H, W = 1080, 1920
ro, rd, vd = get_rays_of_a_view(H, W, [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]], bench.camera(0, "cpu"))
ro, rd, vd = ro.reshape(-1, 3), rd.reshape(-1, 3), vd.reshape(-1, 3)
b = 1000000
o, d, v = ro[b:b + 8192].contiguous(), rd[b:b + 8192].contiguous(), vd[b:b + 8192].contiguous()
for nt in (8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1): break
    torch.set_num_threads(nt)
    model_oracle.fouriergrid_render(state, o, d, v, 1.31)
    t0 = time.perf_counter(); model_oracle.fouriergrid_render(state, o, d, v, 1.31); dt = time.perf_counter() - t0
    print("threads %d: %.2f s/chunk  %.2f Msamples/s" % (nt, dt, 8192 * 256 / dt / 1e6), flush=True)
