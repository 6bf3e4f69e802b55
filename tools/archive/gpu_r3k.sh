#!/bin/bash
mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_gpu_train_scale.py tests/test_gpu_train_step.py tests/test_gpu_touch.py -m gpu -x -q -p no:warnings 2>&1 | tail -25 > gpurun_out/r3k/pytest.log
cat gpurun_out/r3k/pytest.log
: > gpurun_out/r3k/ab.txt
for ph in 1 10001; do for i in 1 2; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph 2>/dev/null | tail -1 >> gpurun_out/r3k/ab.txt
done; done
python - <<'PY'
import json
for l in open('gpurun_out/r3k/ab.txt'):
    d=json.loads(l); print(d['tv_phase'], '%.3f ms'%d['ms_per_step'], {k:round(v,3) for k,v in d['phases_ms'].items()}, d['survivors_M'])
PY
