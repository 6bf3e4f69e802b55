#!/bin/bash
# round-4 visit C: the tests that failed or were new in visit B, the DVGO residual path, a bench line of the new default build.
cd $GRAFT_REPO_ROOT
T=r4c
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_long.py -x -q -s > gpurun_out/$T/pytest_train_long.log 2>&1
timeout 900 python -m pytest tests/test_gpu_reference_callers.py -x -q > gpurun_out/$T/pytest_refcallers.log 2>&1
timeout 900 python -m pytest tests/test_gpu_s1_scale.py -x -q -s -k "fp64 or headline" > gpurun_out/$T/pytest_fp64.log 2>&1
timeout 900 python -m pytest tests/test_dvgo.py tests/test_dcvgo.py -x -q -m gpu > gpurun_out/$T/pytest_dvgo.log 2>&1
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/$T/bench_line.json 2> gpurun_out/$T/bench_err.log
for f in train_long refcallers fp64 dvgo; do echo "== $f"; tail -4 gpurun_out/$T/pytest_$f.log; done
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c/bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("ms_per_step", "value", "frame_sha16")}, {k: round(v["ms"], 3) for k, v in d["kernels"].items()})
print("S1b", d["secondary"]["ms_per_step"], d["secondary"]["kernels"])
print("garden", d.get("secondary_garden_single_sampling"))
s3 = d.get("secondary_s3_train_step", {})
print("S3", {k: s3.get(k) for k in ("ms_per_step", "masked", "roofline_tv_adam_dense")} if isinstance(s3, dict) else s3)
PY
