#!/bin/bash
TAG=${1:-r5g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_touch.py tests/test_gpu_grid_autograd.py -q -p no:warnings -m gpu 2>&1 | tail -5 | tee $OUT/pytest_ops.log
for t in 3 2 3 2; do
  timeout 200 python tools/bench_tv_adam_dense.py --tune tv_xcd=$t 2>/dev/null | tail -1 | tee -a $OUT/tv_adam_dense.jsonl
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
    if "k_tv_cl" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][:30], r["Counter_Name"])].append(float(r["Counter_Value"]))
for c, v in agg.items():
    print("%-30s %-28s n=%d mean=%.6g" % (c[0], c[1], len(v), sum(v) / len(v)))
PY
done
for ph in 1 10001; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph 2>/dev/null | tail -1 >> $OUT/train_step_s3.jsonl
done
python - $OUT/train_step_s3.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['tv_phase'], '%.3f ms' % d['ms_per_step'], {k: round(v, 3) for k, v in d['phases_ms'].items()}, d.get('roofline_tv_adam_dense'))
PY
timeout 600 python tools/bench_voxgo_train.py --model dcvgo 2>/dev/null | cut -c1-330 | tee $OUT/voxgo_train.jsonl
