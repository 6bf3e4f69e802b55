#!/bin/bash
# round 6, visit Q: every training kernel with unconditional loads (corner gathers, TV neighbours, bf16x3 operands): the suite file by file,
# kernel stats of the three training steps, the step clocks
OUT=gpurun_out/r6q; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for f in tests/test_gpu_*.py; do
  b=$(basename $f .py)
  timeout 900 python -X faulthandler -m pytest $f -m gpu -q -x 2>&1 | grep -v "Warning\|warnings.warn\|^$" > $OUT/$b.log
  echo "$b: $(grep -E "passed|failed|error|Fatal|dumped" $OUT/$b.log | tail -2 | tr '\n' ' ')"
done | tee $OUT/summary.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o dvgo -- python $R/tools/bench_voxgo_train.py --model dvgo --steps 40 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_dvgo.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "dvgo_kernel_stats.csv" | head -1) $OUT/voxgo_train_dvgo_kernel_stats.csv 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s3 -- python $R/tools/bench_train_step.py --steps 20 --first-step 10001 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_s3.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "s3_kernel_stats.csv" | head -1) $OUT/train_step_s3_masked_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
python - <<'PY'
import csv
for f in ("voxgo_train_dvgo", "train_step_s3_masked"):
    print("==", f)
    for r in list(csv.DictReader(open("gpurun_out/r6q/%s_kernel_stats.csv" % f)))[:26]:
        if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
        print("%-62s calls %4s avg %8.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for sf in 0 1; do timeout 600 python tools/bench_voxgo_train.py --steps 40 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/voxgo_train.jsonl | cut -c1-60,240-330; done
for sf in 0 1; do timeout 600 python tools/bench_train_step.py --steps 20 --first-step 10001 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/s3_masked.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s3 masked sync_free', d.get('sync_free'), round(d['ms_per_step'],4))"; done
timeout 600 python tools/bench_train_step.py --steps 20 --first-step 1 2>&1 | grep '^{' | tee -a $OUT/s3_dense.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s3 dense', round(d['ms_per_step'],4), d.get('roofline_tv_adam_dense'))"
