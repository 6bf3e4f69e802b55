#!/bin/bash
# fused DCVGO (tests + 1080p timing), march double-angle A/B (time + parity), G=200 train parity
TAG=${1:-r3g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== dcvgo tests"
timeout 900 python -m pytest tests/test_dcvgo.py -m gpu -x -q -s 2>&1 | grep -v "Warning\|warn" | tail -12 | tee $OUT/pytest_dcvgo.log
echo "== dcvgo 1080p"
timeout 900 python tools/bench_dcvgo.py --out $OUT/dcvgo_1080p.json 2>$OUT/dcvgo_err.txt | cut -c1-1800; tail -3 $OUT/dcvgo_err.txt | grep -v amdgpu.ids
echo "== march A/B: product vs double-angle sincos (time, then parity on 131 072 S1 rays)"
for lib in unboundednerfpytorch_amd/libugrid_hip.so build/ab/lib_march_dblangle.so; do
  UGRID_LIB=$lib timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', {k:round(v['ms'],3) for k,v in d['kernels'].items()}, round(d['ms_per_step'],3))" | tee -a $OUT/march_ab.txt
done
for lib in unboundednerfpytorch_amd/libugrid_hip.so build/ab/lib_march_dblangle.so; do
  UGRID_LIB=$lib timeout 400 python tools/gpu_parity_ab.py $(basename $lib .so) 2>&1 | grep -v -i warn | tee -a $OUT/march_ab.txt
done
echo "== train parity at G=200"
timeout 900 python -m pytest tests/test_gpu_train_scale.py -x -q -k "oracle_backend" 2>&1 | tail -4 | tee $OUT/pytest_train200.log
