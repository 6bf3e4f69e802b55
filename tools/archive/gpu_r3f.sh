#!/bin/bash
# full -m gpu suite on the current tree + shade geometry A/B
TAG=${1:-r3f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for pc in 2 1 0; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --tune shade_pc=$pc 2>$OUT/ab_err.txt < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shade_pc=$pc', {k:round(v['ms'],3) for k,v in d['kernels'].items()}, round(d['ms_per_step'],3))" | tee -a $OUT/ab.txt
done
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "Warning\|warn" | tail -25 | tee $OUT/pytest_gpu.log
