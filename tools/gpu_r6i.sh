#!/bin/bash
# round 6, visit I: the whole GPU suite on the current tree, the driver's bench command, rocprofv3 kernel stats of the bench command and of the
# three training steps (sync-free, deferred loss), matrix-pipe counters of the DVGO step's rgbnet kernels
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/r6i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.log
wc -c $OUT/bench_line.json; cp bench_detail.json $OUT/
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-proxy > $R/$OUT/prof_bench.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "s1_kernel_stats.csv" | head -1) $OUT/bench_s1_kernel_stats.csv 2>/dev/null
for m in dvgo dcvgo; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o $m -- python $R/tools/bench_voxgo_train.py --model $m --phase masked --steps 40 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_$m.log 2>&1 < /dev/null )
  cp $(find $OUT/prof -name "${m}_kernel_stats.csv" | head -1) $OUT/voxgo_train_${m}_kernel_stats.csv 2>/dev/null
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s3 -- python $R/tools/bench_train_step.py --steps 20 --first-step 10001 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_s3.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "s3_kernel_stats.csv" | head -1) $OUT/train_step_s3_masked_kernel_stats.csv 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --output-format csv -d /tmp/pmc_dvgo -o p -- python $R/tools/bench_voxgo_train.py --model dvgo --steps 20 --sync-free 1 --lazy-loss 1 > $R/$OUT/pmc_dvgo.log 2>&1 < /dev/null )
f=$(find /tmp/pmc_dvgo -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $OUT/voxgo_train_dvgo_pmc.txt
import sys, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if any(t in k for t in ("k_lin", "k_wgrad", "k_l3", "k_grid_query", "k_train_march")):
        agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per-launch means, rocprofv3 --pmc over tools/bench_voxgo_train.py --model dvgo --sync-free 1 --lazy-loss 1 (matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE))")
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * 256 * m["GRBM_GUI_ACTIVE"]) if m.get("GRBM_GUI_ACTIVE") else float("nan")
    print("%-62s n=%3d GUI_ACTIVE=%9.0f MFMA_BUSY=%12.0f -> matrix pipe busy %.3f  INSTS_VALU=%.4g" % (k, len(next(iter(d.values()))), m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), busy, m.get("SQ_INSTS_VALU", 0)))
PY
rm -rf $OUT/prof
