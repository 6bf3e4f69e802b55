#!/bin/bash
# A/B bench variants in one GPU visit: tools/gpu_ab.sh tag "flags A" "flags B" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for fl in "$@"; do
  i=$((i+1))
  echo "== variant $i: $fl"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $fl 2>&1 | tail -1 | tee $OUT/ab_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('ms/step %.2f  march %.2f ms  shade %.2f ms  value %.0f Msamples/s' % (d['ms_per_step'], k['render_march']['ms'], k['render_shade']['ms'], d['value']))"
done
