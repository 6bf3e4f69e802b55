#!/bin/bash
# round 6, visit F: the sync-free training step -- parity / overflow / hipGraph tests, then the clock (host-counted vs sync-free, eager vs lazy loss)
OUT=gpurun_out/r6f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_voxgo_train.py tests/test_gpu_train_scale.py -x -q -k "sync_free or capturable" 2>&1 | tail -25 | tee $OUT/pytest_sync_free.log
for sf in 0 1; do for lz in 0 1; do
  timeout 600 python tools/bench_voxgo_train.py --steps 40 --sync-free $sf --lazy-loss $lz 2>&1 | grep '^{' | tee -a $OUT/voxgo_train.jsonl
  timeout 600 python tools/bench_train_step.py --steps 20 --first-step 10001 --sync-free $sf --lazy-loss $lz 2>&1 | grep '^{' | tee -a $OUT/s3_masked.jsonl
done; done
timeout 900 python -m pytest tests/test_gpu_train_scale.py tests/test_gpu_voxgo_train.py tests/test_gpu_train_long.py tests/test_gpu_ops.py -x -q 2>&1 | tail -8 | tee $OUT/pytest_train.log
