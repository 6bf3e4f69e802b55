import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from oracle import model_oracle
from test_oracle_golden import make_state
from unboundednerfpytorch_amd import fourier_render as fr
G, F, C, R = 36, 3, 12, 60000
state = make_state(4242, G, F, C, 4, "inf", 1e-4, 6.0, 12.0)
o, d, v = [torch.from_numpy(a) for a in synth.rays(4243, R)]
og, dg, vg = o.cuda(), d.cuda(), v.cuda()
base = None
for mode in (0, 1, 2):
    rend = fr.FourierGridRenderer(state, "cuda:0", mlp_mode=mode)
    outs = []
    for rep in range(12):
        out = rend(og, dg, vg, stepsize=0.5, render_depth=True)
        outs.append(out["rgb_marched"].cpu())
    nd = [int(((outs[0] - x).abs().amax(dim=1) > 0).sum()) for x in outs[1:]]
    if base is None: base = outs[0]
    print("mode", mode, "rays differing from rep0 in 11 reps", nd, "max |mode - fp32| %.2e" % float((outs[0] - base).abs().max()))
