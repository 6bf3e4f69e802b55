#!/bin/bash
# Round-3 first GPU visit: producer/consumer shade A/B, arbitration against the reference on the GPU, S5 two blocks,
# self-launched 2-rank bench (shared GPU), L1 calibration, then the whole -m gpu suite.  Every step under its own timeout.
TAG=${1:-r3a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt; nproc >> $OUT/gpu.txt
echo "== L1 microbench"
timeout 120 hipcc --offload-arch=gfx950 -O3 -o /tmp/l1_dwordx4 tools/microbench/l1_dwordx4.hip 2> $OUT/l1_build.log && timeout 60 /tmp/l1_dwordx4 | tee $OUT/microbench_l1_dwordx4.json
echo "== shade A/B (pc=1 / pc=0), no cpu baseline"
for pc in 1 0; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --tune shade_pc=$pc 2>$OUT/ab_err_$pc.txt < /dev/null | tail -1 > $OUT/ab_pc$pc.json
  python - $OUT/ab_pc$pc.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], {k:round(v["ms"],3) for k,v in d["kernels"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
echo "== targeted tests"
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_s5_blocks.py -x -q 2>&1 | tail -8 | tee $OUT/pytest_fused.log
timeout 1200 python -m pytest tests/test_gpu_s1_scale.py -x -q -s 2>&1 | grep -v Warning | tail -40 | tee $OUT/pytest_s1_scale.log
echo "== bench (full line)"
timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt < /dev/null; tail -c 1500 $OUT/bench_line.json; tail -3 $OUT/bench_err.txt
echo "== bench --gpus 2 self-launched, two gloo ranks on the one GPU"
UGRID_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-chunks 4 > $OUT/bench_2rank_shared_gpu.json 2> $OUT/bench2_err.txt < /dev/null; tail -c 1200 $OUT/bench_2rank_shared_gpu.json; tail -3 $OUT/bench2_err.txt
echo "== S5: two real blocks, shared GPU, check vs single-process rule"
UGRID_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 tools/bench_s5_blocks.py --steps 2 --check --out $OUT/s5_2blocks_shared_gpu.json 2> $OUT/s5_err.txt < /dev/null | tail -1 | cut -c1-900; tail -3 $OUT/s5_err.txt
ls -la $OUT
