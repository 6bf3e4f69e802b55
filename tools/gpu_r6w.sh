#!/bin/bash
# round 6, visit W: the touched-line walker with up to 64 words per wave -- bit-equality tests, then S3's masked step
OUT=gpurun_out/r6w; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_touch.py tests/test_gpu_train_scale.py tests/test_gpu_voxgo_train.py tests/test_gpu_train_long.py tests/test_gpu_multi.py -q -x 2>&1 | tail -3 | tee $OUT/pytest.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s3 -- python $R/tools/bench_train_step.py --steps 20 --first-step 10001 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_s3.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "s3_kernel_stats.csv" | head -1) $OUT/train_step_s3_masked_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r6w/train_step_s3_masked_kernel_stats.csv")))[:14]:
    if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
    print("%-62s calls %4s avg %8.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for sf in 0 1; do timeout 600 python tools/bench_train_step.py --steps 20 --first-step 10001 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/s3_masked.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s3 masked sync_free', d.get('sync_free'), round(d['ms_per_step'],4))"; done
for sf in 0 1; do timeout 600 python tools/bench_voxgo_train.py --steps 40 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/voxgo_train.jsonl | cut -c1-60,240-330; done
