"""Host-side profile (cProfile) of the S3 train step: where the Python / launch time goes.  Prints the top entries by
cumulative and by own time."""
import cProfile
import io
import pstats
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import bench_train_step as bts  # noqa: E402
from unboundednerfpytorch_amd import train_step as ts  # noqa: E402
from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = bts.make_model(200, 4, dev, True, True)
    opt = create_optimizer_or_freeze_model(model, bts.TRUCK_CFG, global_step=0)
    rk = dict(stepsize=0.5, rand_bkgd=True)
    rays = [bts.random_rays(4096, dev, seed=s) for s in range(1, 6)]
    for s in range(1, 4):
        ts.train_iteration(model, opt, *rays[s % 5], bts.TRUCK_CFG, s, rk)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    n = 20
    for s in range(4, 4 + n):
        ts.train_iteration(model, opt, *rays[s % 5], bts.TRUCK_CFG, s, rk)
    torch.cuda.synchronize()
    pr.disable()
    for key in ("cumulative", "tottime"):
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(28)
        txt = buf.getvalue().replace("/root/repo/", "")
        print("\n".join(l[:170] for l in txt.splitlines()[4:]))
    print("steps", n)


if __name__ == "__main__":
    main()
