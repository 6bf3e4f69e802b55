#!/bin/bash
# Round-5 visit R: kernel trace of the native DVGO / DCVGO-masked training steps
OUT=gpurun_out/r5r; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in dvgo dcvgo; do
  rm -rf /tmp/prof_$m
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -o t -- python $R/tools/bench_voxgo_train.py --model $m --phase masked --steps 40 --warmup 4 > $R/$OUT/log_$m.txt 2>&1 )
  f=$(find /tmp/prof_$m -name "t_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/voxgo_train_${m}_kernel_stats.csv
  python - $OUT/voxgo_train_${m}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:40]:
    n = int(r["Calls"]); t = float(r["TotalDurationNs"])
    if n >= 40:
        print("%-70s calls %5d  avg %8.1f us  per step %7.1f us" % (r["Name"][:70], n, float(r["AverageNs"]) / 1e3, t / 44 / 1e3))
        tot += t / 44 / 1e3
print("sum per step (kernels with >= 40 calls): %.1f us" % tot)
PY
done
