#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, kernel-trace profile.  Logs under gpurun_out/.
# usage: tools/gpu_check.sh [tag]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"
timeout 1200 python bench.py --steps 5 --warmup 2 2>&1 | tail -3 | tee $OUT/bench.log
echo "== rocprofv3 kernel trace"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o s1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
rm -f $(find $OUT/prof -name "*.db") $(find $OUT/prof -name "*kernel_trace.csv")
echo "== bench through torch.distributed (1 rank over RCCL: the N>1 code path incl. the overlapped all-gather)"
UGRID_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_dist1.log | cut -c1-400
echo "== training ops"
timeout 300 python tools/bench_train_ops.py 2>&1 | tail -6 | tee $OUT/train_ops.txt
echo "== parity probe"
timeout 300 python tools/gpu_parity_probe.py 2>&1 | grep -v Warning | tail -22 > $OUT/parity_probe.txt; head -3 $OUT/parity_probe.txt
