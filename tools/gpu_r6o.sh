#!/bin/bash
# round 6, visit O: hipGraph bisect (which eager run breaks the captured backward?) + k_lin_b3 with LDS-transposed coalesced row loads
OUT=gpurun_out/r6o; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 800 python tools/dbg_graph_step.py 2>&1 | tee $OUT/dbg_graph.log | grep "^==\|replayed\|Fatal\|captured\|ended"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "rgbnet" 2>&1 | tail -3 | tee $OUT/pytest_rgbnet.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o dvgo -- python $R/tools/bench_voxgo_train.py --model dvgo --steps 40 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_dvgo.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "dvgo_kernel_stats.csv" | head -1) $OUT/voxgo_train_dvgo_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r6o/voxgo_train_dvgo_kernel_stats.csv")):
    if any(t in r["Name"] for t in ("k_lin", "k_wgrad", "k_l3")): print("%-60s calls %4s avg %7.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for sf in 0 1; do timeout 600 python tools/bench_voxgo_train.py --steps 40 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/voxgo_train.jsonl | cut -c1-60,240-330; done
timeout 900 python -m pytest tests/test_gpu_train_scale.py tests/test_gpu_train_long.py tests/test_gpu_voxgo_train.py -q --deselect tests/test_gpu_voxgo_train.py::test_train_step_is_capturable_in_a_hip_graph 2>&1 | tail -3 | tee $OUT/pytest_train.log
