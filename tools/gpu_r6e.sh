#!/bin/bash
# round 6, visit E: the bf16x3 training-rgbnet kernels -- parity tests, then the three training steps with train_mlp = 0 / 1
OUT=gpurun_out/r6e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "rgbnet" 2>&1 | tail -15 | tee $OUT/pytest_rgbnet.log
for m in 0 1; do
  UGRID_TUNE=train_mlp=$m timeout 600 python tools/bench_voxgo_train.py --steps 40 2>&1 | grep -v Warning | tail -4 | tee $OUT/voxgo_train_mlp$m.jsonl
done
timeout 900 python -m pytest tests/test_gpu_train_scale.py tests/test_gpu_voxgo_train.py tests/test_gpu_train_long.py -x -q 2>&1 | tail -8 | tee $OUT/pytest_train.log
