"""Lane efficiency of the march kernel on the bench frame (needs the -DUG_MARCH_STATS build: UGRID_LIB=build/ab/lib_mstats.so).
Prints, per ray order, active lane-iterations / (64 x wave iterations) and wave iterations / (waves x S)."""
import ctypes
import json
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from unboundednerfpytorch_amd import _lib  # noqa: E402


def main():
    L = _lib.load()
    dev = torch.device("cuda", 0)
    res = {}
    for tile in (0, 8):
        sys.argv = ["bench.py", "--ray-tile", str(tile), "--no-cpu-baseline", "--no-secondary"]
        args = bench.parse()
        make = {"s1": bench.make_state, "s1b": bench.make_state_surfaces}
        state = make[args.scene](args.grid, dev, seed=0) if not hasattr(main, "st") else main.st
        main.st = state
        fb = bench.FrameBench(args, state, dev, 1, 0, None)
        fb.step()
        torch.cuda.synchronize()
        h = (ctypes.c_ulonglong * 4)()
        L.ugx_march_stats_read(h)
        fb.step()
        torch.cuda.synchronize()
        L.ugx_march_stats_read(h)
        it, act, s_tot = int(h[0]), int(h[1]), int(h[2])
        res["tile%d" % tile] = {"lane_efficiency": act / (64.0 * it), "iterations_over_S": it / float(s_tot),
                                "active_fraction_of_all_samples": act / (64.0 * s_tot)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
