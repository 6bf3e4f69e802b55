#!/bin/bash
# round 3 visit j: fused rgbnet training kernels
mkdir -p gpurun_out/r3j
timeout 300 python tools/dbg_mlp.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_touch.py tests/test_gpu_train_step.py tests/test_gpu_train_scale.py -m gpu -x -q -p no:warnings 2>&1 | tail -15 > gpurun_out/r3j/pytest.log
cat gpurun_out/r3j/pytest.log
: > gpurun_out/r3j/train_step_ab.txt
for ph in 1 10001; do for t in 1 1; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph --touch $t 2>/dev/null | tail -1 >> gpurun_out/r3j/train_step_ab.txt
done; done
python - <<'PY'
import json
for l in open('gpurun_out/r3j/train_step_ab.txt'):
    d=json.loads(l); print(d['tv_phase'], 'touch', d['touch_bitmap'], '%.3f ms'%d['ms_per_step'], {k:round(v,3) for k,v in d['phases_ms'].items()}, d.get('k0_grad_lines_touched_frac'))
PY
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o p -- python $R/tools/bench_train_step.py --steps 20 --first-step 10001 > /tmp/log.txt 2>&1 < /dev/null
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py "$f" --step 15 > $R/gpurun_out/r3j/masked_step_timeline2.txt
grep "k_lin\|k_wgrad\|^step" $R/gpurun_out/r3j/masked_step_timeline2.txt | cut -c1-90
