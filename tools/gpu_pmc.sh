#!/bin/bash
# PMC passes (each its own rocprofv3 run, kernel-trace only, as the guide prescribes) on the bench workload.
# usage: tools/gpu_pmc.sh tag "CTR1 CTR2" ["CTR3" ...]   -> gpurun_out/<tag>/pmc_<i>/
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 60 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$i -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $BENCH_FLAGS > $GRAFT_REPO_ROOT/$OUT/pmc_$i.log 2>&1 )
  f=$(find $OUT/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "?")
    if "k_march" in k or "k_shade" in k:
        agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in d.items():
        print("%-62s %-28s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
    # keep the per-dispatch csv small: only our kernels
    python - "$f" <<'PY'
import sys
f = sys.argv[1]
lines = open(f).read().splitlines()
keep = [lines[0]] + [l for l in lines[1:] if "k_march" in l or "k_shade" in l]
open(f, "w").write("\n".join(keep) + "\n")
PY
  else
    tail -5 $OUT/pmc_$i.log
  fi
  rm -f $(find $OUT/pmc_$i -name "*.db") $(find $OUT/pmc_$i -name "*kernel_trace.csv") 2>/dev/null
done
