import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from test_oracle_golden import make_state
from unboundednerfpytorch_amd import fourier_render as fr
G, F, C, R = 32, 3, 12, 60000
state = make_state(99, G, F, C, 4, "inf", 1e-4, 6.0, 12.0)
o, d, v = [torch.from_numpy(a).cuda() for a in synth.rays(5, R)]
for mode in (0, 1):
    split = fr.FourierGridRenderer(state, "cuda:0", fused=False, mlp_mode=mode)
    single = fr.FourierGridRenderer(state, "cuda:0", fused=True, mlp_mode=mode)
    a = split(o, d, v, stepsize=0.5, render_depth=True)
    for rep in range(3):
        b = single(o, d, v, stepsize=0.5, render_depth=True)
        for k in ("rgb_marched", "depth", "alphainv_last"):
            diff = (a[k] - b[k]).abs()
            diff = diff.amax(dim=1) if diff.dim() == 2 else diff
            print("mode", mode, "rep", rep, k, "rays differing", int((diff > 0).sum()), "max %.3e" % float(diff.max()))
