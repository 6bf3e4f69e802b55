"""Pick (mu, sigma) of the white-noise density grid of the S1 bench scene so that about 5% of the samples
survive both thresholds.  CPU only; uses the oracle on a strided sample of the frame.
    python tools/calibrate_s1.py MU SIGMA [G]
"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from oracle import model_oracle  # noqa: E402
from unboundednerfpytorch_amd.fourier_render import get_rays_of_a_view  # noqa: E402

mu, sigma = float(sys.argv[1]), float(sys.argv[2])
G = int(sys.argv[3]) if len(sys.argv) > 3 else 200
bench.DENS_MEAN, bench.DENS_STD = mu, sigma
torch.manual_seed(0)
t0 = time.time()
state = bench.make_state(G, "cpu", 0)
print("state built %.1fs" % (time.time() - t0))
H, W = 1080, 1920
K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
ro, rd, vd = get_rays_of_a_view(H, W, K, bench.camera(0, "cpu"))
ro, rd, vd = ro.reshape(-1, 3), rd.reshape(-1, 3), vd.reshape(-1, 3)
idx = torch.arange(0, ro.shape[0], ro.shape[0] // 4096)[:4096]
stepsize = 1.31
t0 = time.time()
out = model_oracle.fouriergrid_render(state, ro[idx], rd[idx], vd[idx], stepsize, render_depth=True)
dt = time.time() - t0
R, S = idx.numel(), out["n_max"]
M = out["weights"].numel()
print("mu=%g sigma=%g: S=%d survivors f=%.4f terminated=%.3f rgb mean=%.3f max=%.3f  (%.1fs, %.2f Msamples/s)" % (
    mu, sigma, S, M / (R * S), float((out["alphainv_last"] < 1e-3).float().mean()),
    float(out["rgb_marched"].mean()), float(out["rgb_marched"].max()), dt, R * S / dt / 1e6))
