"""Freeze outputs of the REFERENCE'S OWN native kernels called with DOUBLE tensors (their AT_DISPATCH_FLOATING_TYPES double
instantiation; oracle/_ref built from FourierGrid/cuda/*.cu by oracle/build_ref.py) into tests/golden/native_ops_f64.npz.
Needs an MI355X:

    gpurun -- python tests/golden/gen_native_golden_f64.py            # writes gpurun_out/native_ops_f64.npz
    cp gpurun_out/native_ops_f64.npz tests/golden/native_ops_f64.npz

Same seeded cases as gen_native_golden.py (tests/native_cases.py, all 18 exported functions), floating inputs cast to float64;
the outputs of the `nofma` build are stored, the `fma` build's are compared (where they differ the reference itself is ambiguous).
The same run prints how the fp64 twins of the HIP library (include/ugrid_hip_f64.h) compare."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import native_cases as nc  # noqa: E402
from oracle import build_ref  # noqa: E402


def main():
    assert torch.cuda.is_available(), "needs the GPU box"
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    res = nc.run_all(build_ref.load("nofma"), scale=1, device="cuda", dtype=torch.float64)
    store = {"%s__%d" % (name, k): o.numpy() for name, outs in res.items() for k, o in enumerate(outs)}
    differs = {}
    if build_ref.built("fma"):
        fma = nc.run_all(build_ref.load("fma"), scale=1, device="cuda", chain_from=res, dtype=torch.float64)
        for name in res:
            for k, (a, b) in enumerate(zip(res[name], fma[name])):
                differs["%s__%d" % (name, k)] = int((a != b).sum()) if not torch.equal(a, b) else 0
    store["report_json"] = np.frombuffer(json.dumps(differs).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(out_dir, "native_ops_f64.npz"), **store)
    from unboundednerfpytorch_amd import adam_upd_cuda, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    hip = {nc.RU: render_utils_cuda, nc.TV: total_variation_cuda, nc.UB: ub360_utils_cuda, nc.AD: adam_upd_cuda}
    got = nc.run_all(hip, scale=1, device="cuda", chain_from=res, dtype=torch.float64)
    bad = [(n, k) for n in res for k, (a, b) in enumerate(zip(res[n], got[n])) if not np.array_equal(a.numpy(), b.numpy(), equal_nan=True)]
    print("fma build differs from nofma in:", {k: v for k, v in differs.items() if v})
    print("HIP fp64 twins differ from the reference (nofma) in:", bad)
    print("wrote", os.path.join(out_dir, "native_ops_f64.npz"), os.path.getsize(os.path.join(out_dir, "native_ops_f64.npz")), "bytes")


if __name__ == "__main__":
    main()
