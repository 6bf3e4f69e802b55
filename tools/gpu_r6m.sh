#!/bin/bash
# round 6, visit M: (1) two-phase scalar-uniform march arms; (2) hipGraph bisect with the test's own sequence; (3) wave-state counters of k_lin_b3
OUT=gpurun_out/r6m; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
AB_NAME=s1 AB_REPS=2 BENCH_FLAGS="--no-truck --no-proxy" bash tools/gpu_ab.sh r6m build/ab/base6.so build/ab/su_g1.so
AB_NAME=s1w5 AB_REPS=1 BENCH_FLAGS="--no-truck --no-proxy --tune march_waves=5" bash tools/gpu_ab.sh r6m build/ab/base6.so build/ab/su_g1.so build/ab/su_g2.so
AB_NAME=s1w4 AB_REPS=1 BENCH_FLAGS="--no-truck --no-proxy --tune march_waves=4" bash tools/gpu_ab.sh r6m build/ab/base6.so build/ab/su_g2.so build/ab/su_g3.so
AB_NAME=truck AB_REPS=1 AB_STEPS=8 BENCH_FLAGS="--no-truck --no-proxy --scene s1b --freq 4 --stepsize 0.5" bash tools/gpu_ab.sh r6m build/ab/base6.so build/ab/su_g1.so
timeout 900 python tools/dbg_graph_step.py 2>&1 | tee $OUT/dbg_graph.log | grep "^==\|replayed\|Fatal" 
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_$i -o p -- python $R/tools/bench_voxgo_train.py --model dvgo --steps 12 --sync-free 1 --lazy-loss 1 > $R/$OUT/pmc_$i.log 2>&1 < /dev/null )
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee -a $OUT/klin_b3_pmc.txt
import sys, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_lin_b3<8" in k or "k_wgrad_b3" in k or "k_lin<64, 1" in k:
        agg[k.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
