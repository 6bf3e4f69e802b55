#!/bin/bash
# round 6, visit S: k_lin_b3 with the hand-pipelined k-step loop
OUT=gpurun_out/r6s; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -X faulthandler -m pytest tests/test_gpu_ops.py -q -x -k "rgbnet" 2>&1 | tail -4 | tee $OUT/pytest_rgbnet.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o dvgo -- python $R/tools/bench_voxgo_train.py --model dvgo --steps 40 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_dvgo.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "dvgo_kernel_stats.csv" | head -1) $OUT/voxgo_train_dvgo_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r6s/voxgo_train_dvgo_kernel_stats.csv")):
    if any(t in r["Name"] for t in ("k_lin", "k_wgrad", "k_l3")): print("%-60s calls %4s avg %7.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for sf in 0 1; do timeout 600 python tools/bench_voxgo_train.py --steps 40 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/voxgo_train.jsonl | cut -c1-60,240-330; done
timeout 900 python -m pytest tests/test_gpu_train_scale.py tests/test_gpu_train_long.py tests/test_gpu_voxgo_train.py -q 2>&1 | tail -3 | tee $OUT/pytest_train.log
