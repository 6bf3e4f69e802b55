#!/bin/bash
# Round-5 visit T: kernel trace of the S3 training step (masked-TV phase), native step
OUT=gpurun_out/r5t; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_s3
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s3 -o t -- python $R/tools/bench_train_step.py --steps 40 --first-step 10001 > $R/$OUT/log_s3.txt 2>&1 )
f=$(find /tmp/prof_s3 -name "t_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/train_step_s3_masked_kernel_stats.csv
python - $OUT/train_step_s3_masked_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:45]:
    n = int(r["Calls"]); t = float(r["TotalDurationNs"])
    if n >= 40:
        print("%-70s calls %5d  avg %8.1f us" % (r["Name"][:70], n, float(r["AverageNs"]) / 1e3))
PY
tail -1 $OUT/log_s3.txt | cut -c1-300
