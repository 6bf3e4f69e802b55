#!/bin/bash
# A/B bench variants in one GPU visit: tools/gpu_ab.sh tag "flags A" "ENV=x|flags B" ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
i=0
for v in "$@"; do
  i=$((i+1))
  if [[ "$v" == *"|"* ]]; then envs="${v%%|*}"; fl="${v#*|}"; else envs=""; fl="$v"; fi
  echo "== variant $i: $v"
  env $envs timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $fl 2>&1 | tail -1 | tee $OUT/ab_$i.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('ms/step %.2f  %s  value %.0f Msamples/s' % (d['ms_per_step'], '  '.join('%s %.2f ms' % (n, v['ms']) for n, v in k.items()), d['value']))"
done
