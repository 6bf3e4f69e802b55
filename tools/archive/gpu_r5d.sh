#!/bin/bash
# round 5, visit D: host-stall diagnostic of the rank shares, multi-tensor Adam tests, training-step benches after the host-side diet
TAG=${1:-r5d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python tools/diag_share_stall.py 2>&1 | grep -v "amdgpu.ids" | tail -40 | tee $OUT/diag_share_stall.txt
timeout 600 python -m pytest tests/test_gpu_adam_multi.py tests/test_gpu_ops.py -q -p no:warnings -m gpu 2>&1 | tail -5 | tee $OUT/pytest_adam.log
timeout 600 python tools/bench_voxgo_train.py --model both > $OUT/voxgo_train.jsonl 2>/dev/null; cut -c1-260 $OUT/voxgo_train.jsonl
for ph in 1 10001; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph 2>/dev/null | tail -1 >> $OUT/train_step_s3.jsonl
done
python - $OUT/train_step_s3.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['tv_phase'], '%.3f ms' % d['ms_per_step'], {k: round(v, 3) for k, v in d['phases_ms'].items()}, d.get('roofline_tv_adam_dense'))
PY
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_voxgo_train.py tests/test_gpu_train_scale.py -q -p no:warnings -m gpu -x 2>&1 | tail -5 | tee $OUT/pytest_train.log
ls $OUT
