"""Freeze outputs of the REFERENCE'S OWN native kernels (oracle/_ref, built from FourierGrid/cuda/*.cu by
oracle/build_ref.py) into tests/golden/native_ops.npz.  Needs an MI355X:

    python oracle/build_ref.py                                        # build container (has /root/reference)
    gpurun -- python tests/golden/gen_native_golden.py                # writes gpurun_out/native_ops.npz (+ .txt log)
    cp gpurun_out/native_ops.npz tests/golden/native_ops.npz

Inputs are the seeded cases of tests/native_cases.py (all 18 exported functions).  Stored: the outputs of the
`nofma` build (-ffp-contract=off, the semantics oracle/ref_ops.c restates), plus, per output, the largest ulp distance
of the `fma` build (hipcc's default contraction ~ nvcc's default -fmad=true) -- where that is non-zero the reference
itself is ambiguous at that level.  The same run prints how the HIP library and the C oracle compare, as a first check.
"""
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
    ref = build_ref.load("nofma")
    res = nc.run_all(ref, scale=1, device="cuda")
    store = {}
    for name, outs in res.items():
        for k, o in enumerate(outs):
            store["%s__%d" % (name, k)] = o.numpy()
    report = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0), "fma_vs_nofma_max_ulp": {},
              "hip_vs_ref": {}, "c_oracle_vs_ref": {}}
    if build_ref.built("fma"):
        fma = nc.run_all(build_ref.load("fma"), scale=1, device="cuda", chain_from=res)
        for name in res:
            for k, (a, b) in enumerate(zip(res[name], fma[name])):
                if a.dtype == torch.float32:
                    fin = np.isfinite(a.numpy()) & np.isfinite(b.numpy())
                    same_nonfinite = np.array_equal(np.isfinite(a.numpy()), np.isfinite(b.numpy()))
                    d = int(nc.ulp_diff(a.numpy()[fin], b.numpy()[fin]).max()) if fin.any() else 0
                    report["fma_vs_nofma_max_ulp"]["%s__%d" % (name, k)] = d if same_nonfinite else -1
                else:
                    report["fma_vs_nofma_max_ulp"]["%s__%d" % (name, k)] = 0 if torch.equal(a, b) else -1
    store["report_json"] = np.frombuffer(json.dumps(report["fma_vs_nofma_max_ulp"]).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(out_dir, "native_ops.npz"), **store)

    # first look: HIP library and C oracle against the reference kernels on the same inputs
    from oracle import ref_ops
    from unboundednerfpytorch_amd import adam_upd_cuda, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    hip = {nc.RU: render_utils_cuda, nc.TV: total_variation_cuda, nc.UB: ub360_utils_cuda, nc.AD: adam_upd_cuda}
    orc = {nc.RU: ref_ops.render_utils_cuda, nc.TV: ref_ops.total_variation_cuda, nc.UB: ref_ops.ub360_utils_cuda,
           nc.AD: ref_ops.adam_upd_cuda}
    got_hip = nc.run_all(hip, scale=1, device="cuda", chain_from=res)
    got_orc = nc.run_all(orc, scale=1, device=None, chain_from=res)
    for label, got in (("hip_vs_ref", got_hip), ("c_oracle_vs_ref", got_orc)):
        for name in res:
            for k, (a, b) in enumerate(zip(res[name], got[name])):
                key = "%s__%d" % (name, k)
                if a.shape != b.shape or a.dtype != b.dtype:
                    report[label][key] = "shape/dtype %s %s vs %s %s" % (tuple(a.shape), a.dtype, tuple(b.shape), b.dtype)
                elif a.dtype == torch.float32:
                    an, bn = a.numpy(), b.numpy()
                    fin = np.isfinite(an) & np.isfinite(bn)
                    nf_same = np.array_equal(np.isnan(an), np.isnan(bn)) and np.array_equal(an[~fin & ~np.isnan(an)], bn[~fin & ~np.isnan(an)])
                    report[label][key] = {"max_ulp": int(nc.ulp_diff(an[fin], bn[fin]).max()) if fin.any() else 0,
                                          "max_abs": float(np.abs(an[fin] - bn[fin]).max()) if fin.any() else 0.0,
                                          "nonfinite_same": bool(nf_same)}
                else:
                    report[label][key] = {"equal": bool(torch.equal(a, b))}
    txt = json.dumps(report, indent=1)
    open(os.path.join(out_dir, "native_ops_report.json"), "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
