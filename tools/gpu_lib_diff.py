"""Compare the render of several builds of libugrid_hip.so (A/B debugging): python tools/gpu_lib_diff.py libA.so libB.so ...
Each library is loaded in its own subprocess (UGRID_LIB), outputs are compared against the first."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch, numpy as np
ROOT = sys.argv[1]; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from test_oracle_golden import make_state
from unboundednerfpytorch_amd import fourier_render as fr
state = make_state(4242, 36, 3, 12, 4, "inf", 1e-4, 6.0, 12.0)
o, d, v = [torch.from_numpy(a).cuda() for a in synth.rays(4243, 6000)]
res = {}
for mode in (0, 2):
    out = fr.FourierGridRenderer(state, "cuda:0", mlp_mode=mode)(o, d, v, stepsize=0.5, render_depth=True)
    res["rgb%d" % mode] = out["rgb_marched"].cpu().numpy()
np.savez(sys.argv[2], **res)
'''
outs = []
for lib in sys.argv[1:]:
    f = tempfile.mktemp(suffix=".npz")
    env = dict(os.environ, UGRID_LIB=os.path.abspath(lib))
    subprocess.check_call([sys.executable, "-c", CHILD, ROOT, f], env=env)
    outs.append(dict(np.load(f)))
for lib, o in zip(sys.argv[2:], outs[1:]):
    for k in o:
        dlt = np.abs(o[k] - outs[0][k]).max(axis=1)
        bad = np.nonzero(dlt > 1e-5)[0]
        print(lib, k, "rays differing >1e-5: %d of %d, max %.3e" % (len(bad), len(dlt), dlt.max()), "first:", bad[:12].tolist(),
              "tile hist:", np.bincount(bad % 64, minlength=64)[:16].tolist() if len(bad) else "")
        if len(bad):
            i = bad[0]
            print("   ray", i, "A", outs[0][k][i], "B", o[k][i])
