#!/bin/bash
# round 6, visit R: sin / cos level pairs in the training gathers (16 loads in flight); suite file by file; S3 / DVGO kernel stats; wave-state
# counters of the bf16x3 kernels
OUT=gpurun_out/r6r; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for f in tests/test_gpu_grid_autograd.py tests/test_gpu_ops.py tests/test_gpu_ref_native.py tests/test_gpu_reference_callers.py tests/test_gpu_touch.py tests/test_gpu_train_step.py tests/test_gpu_train_scale.py tests/test_gpu_voxgo_train.py tests/test_gpu_train_long.py tests/test_fourier_model.py tests/test_dvgo.py tests/test_dcvgo.py; do
  b=$(basename $f .py)
  timeout 900 python -X faulthandler -m pytest $f -m gpu -q -x 2>&1 | grep -v "Warning\|warnings.warn\|^$" > $OUT/$b.log
  echo "$b: $(grep -E "passed|failed|error|Fatal|dumped|no tests" $OUT/$b.log | tail -2 | tr '\n' ' ')"
done | tee $OUT/summary.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s3 -- python $R/tools/bench_train_step.py --steps 20 --first-step 10001 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_s3.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "s3_kernel_stats.csv" | head -1) $OUT/train_step_s3_masked_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r6r/train_step_s3_masked_kernel_stats.csv")))[:22]:
    if "at::native" in r["Name"] or "rocclr" in r["Name"]: continue
    print("%-62s calls %4s avg %8.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
for sf in 0 1; do timeout 600 python tools/bench_train_step.py --steps 20 --first-step 10001 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/s3_masked.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s3 masked sync_free', d.get('sync_free'), round(d['ms_per_step'],4))"; done
for sf in 0 1; do timeout 600 python tools/bench_voxgo_train.py --steps 40 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/voxgo_train.jsonl | cut -c1-60,240-330; done
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT" "TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$i -o p -- python $R/tools/bench_voxgo_train.py --model dvgo --steps 12 --sync-free 1 --lazy-loss 1 > $R/$OUT/pmc_$i.log 2>&1 < /dev/null )
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee -a $OUT/klin_b3_pmc.txt
import sys, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_lin_b3" in k or "k_wgrad_b3" in k:
        agg[k.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
