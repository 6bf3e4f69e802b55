#!/bin/bash
# shade kernel iteration: phase profiles of the A/B libraries under build/ab/*prof*.so, then an A/B of the product library
TAG=${1:-r3c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for pc in 2 1 0 2; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --tune shade_pc=$pc 2>$OUT/ab_err.txt < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shade_pc=$pc', {k:round(v['ms'],3) for k,v in d['kernels'].items()}, round(d['ms_per_step'],3))" | tee -a $OUT/ab.txt
done
tail -3 $OUT/ab_err.txt | grep -v amdgpu.ids
for lib in build/ab/*prof*.so; do
  n=$(basename $lib .so)
  UGRID_LIB=$lib timeout 300 python tools/gpu_shade_pc_prof.py s1 2>$OUT/err_$n.txt | tee $OUT/phases_$n.txt; tail -2 $OUT/err_$n.txt | grep -v amdgpu.ids
done
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -3 | tee $OUT/pytest_fused.log
