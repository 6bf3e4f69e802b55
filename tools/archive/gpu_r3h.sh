#!/bin/bash
# dual-pass consumer: bit identity + A/B timing
TAG=${1:-r3h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -k "geometries or deterministic" 2>&1 | tail -6 | tee $OUT/pytest_geo.log
for pc in 3 2 0 3; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --tune shade_pc=$pc 2>$OUT/ab_err.txt < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shade_pc=$pc', {k:round(v['ms'],3) for k,v in d['kernels'].items()}, round(d['ms_per_step'],3))" | tee -a $OUT/ab.txt
done
tail -3 $OUT/ab_err.txt | grep -v amdgpu.ids
