#!/bin/bash
# PMC passes (each its own rocprofv3 run, --kernel-trace only, as MI355X_MICROARCH.md prescribes) on the bench workload.
#   usage: tools/gpu_pmc.sh <tag> "CTR1 CTR2" ["CTR3" ...]   -> gpurun_out/<tag>/pmc_<i>/, pmc_csv/pmc_pass_<i>.csv
# Standard groups of round 4 (tools/gpu_pmc.sh r4x $PMC_STD): see PMC_GROUPS in tools/pmc_summarize.py
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT/pmc_csv
export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$i -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --frame-pair 0 --no-cpu-baseline --no-secondary --no-proxy $BENCH_FLAGS > $GRAFT_REPO_ROOT/$OUT/pmc_$i.log 2>&1 )
  f=$(find $OUT/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    # keep the per-dispatch csv small: only our kernels
    python - "$f" $OUT/pmc_csv/pmc_pass_$i.csv <<'PY'
import sys, csv, collections
lines = open(sys.argv[1]).read().splitlines()
keep = [lines[0]] + [l for l in lines[1:] if "k_march" in l or "k_shade" in l]
open(sys.argv[2], "w").write("\n".join(keep) + "\n")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(keep):
    agg[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in d.items():
        print("%-50s %-34s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
  else
    echo "pass $i ($ctrs): no counter csv"; tail -5 $OUT/pmc_$i.log
  fi
  rm -rf $OUT/pmc_$i
done
