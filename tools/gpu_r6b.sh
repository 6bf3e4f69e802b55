#!/bin/bash
# round 6, visit B: the Fourier loss inside RenderLoss / the native step + a 100-repetition spread
OUT=gpurun_out/r6b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_scale.py tests/test_gpu_voxgo_train.py tests/test_gpu_train_long.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_train.log
timeout 1200 python tools/native_step_spread.py --reps 100 --out $OUT/native_step_spread.json 2>&1 | grep -v Warning | tail -6
