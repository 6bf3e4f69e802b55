#!/bin/bash
# Round-3 second GPU visit: where the producer / consumer shade kernel spends its time (+ A/B libraries), the arbitration
# tests, the full bench line, the self-launched 2-rank bench, the MFMA hazard reproducer.
TAG=${1:-r3b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pc profile"
UGRID_LIB=build/ab/lib_pc_prof.so timeout 300 python tools/gpu_shade_pc_prof.py s1 2>$OUT/prof_err.txt | tee $OUT/shade_pc_phases_s1.txt; tail -2 $OUT/prof_err.txt
echo "== A/B libraries"
for lib in build/ab/lib_pc_nbl4.so build/ab/lib_pc_slots2.so; do
  UGRID_LIB=$lib timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', {k:round(v['ms'],3) for k,v in d['kernels'].items()}, round(d['ms_per_step'],3))" | tee -a $OUT/ab_libs.txt
done
echo "== MFMA hazard reproducer"
timeout 120 hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dep_hazard tools/microbench/mfma_dep_hazard.hip 2>/dev/null && timeout 120 /tmp/mfma_dep_hazard | tee $OUT/mfma_dep_hazard.json
echo "== arbitration"
timeout 1200 python -m pytest tests/test_gpu_s1_scale.py -x -q -s -k "arbitration or headline" 2>&1 | grep -v Warning | tail -30 | tee $OUT/pytest_arbitration.log
echo "== bench (full line)"
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt < /dev/null; tail -c 2500 $OUT/bench_line.json; echo; tail -3 $OUT/bench_err.txt
echo "== bench --gpus 2 self-launched, two gloo ranks on the one GPU"
UGRID_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-chunks 4 > $OUT/bench_2rank_shared_gpu.json 2> $OUT/bench2_err.txt < /dev/null; tail -c 1500 $OUT/bench_2rank_shared_gpu.json; echo; grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*" $OUT/bench2_err.txt | tail -5
ls -la $OUT
