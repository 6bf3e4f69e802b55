#!/bin/bash
# round 5, visit F: the dense TV + Adam pass alone -- time per tv_xcd mode, and the HBM bytes it really moves (TCC request counters)
TAG=${1:-r5f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for lib in "" $EXTRA_LIBS; do
  for t in 2 1 0; do
    [ -n "$lib" ] && [ $t != 2 ] && continue
    UGRID_LIB=$lib timeout 200 python tools/bench_tv_adam_dense.py --tune tv_xcd=$t 2>/dev/null | tail -1 | sed "s|^|lib=$lib |" | tee -a $OUT/tv_adam_dense.jsonl
  done
done
i=0
for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/tv_$i -o p -- python $R/tools/bench_tv_adam_dense.py --reps 4 > $R/$OUT/tv_pmc_$i.log 2>&1 )
  f=$(find /tmp/tv_$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee -a $OUT/tv_adam_dense_pmc.txt
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_tv_cl_vec4" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items():
    print("%-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
done
ls $OUT
