#!/bin/bash
# Round-5 visit S: the native step for FourierGridModel (mode 'fourier') -- parity, then the S3 / voxgo clocks
OUT=gpurun_out/r5s; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_train_scale.py tests/test_gpu_voxgo_train.py tests/test_gpu_train_step.py tests/test_gpu_touch.py -m gpu -q -p no:warnings -x 2>&1 | tail -12
for ph in 1 10001; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['tv_phase'], round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['phases_ms'].items()})"
done
timeout 300 python tools/bench_voxgo_train.py --model both --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['model'], d['workload'][-12:], round(d['ms_per_step'], 4))
"
