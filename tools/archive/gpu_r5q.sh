#!/bin/bash
# Round-5 visit Q: the native training step (native_step.VoxGOStep) -- parity with the op-by-op step, then the clock
OUT=gpurun_out/r5q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_voxgo_train.py -m gpu -q -p no:warnings -x 2>&1 | tail -15
for l in 0 1; do
  timeout 600 python tools/bench_voxgo_train.py --model both --lazy-loss $l > $OUT/voxgo_train_lazy$l.jsonl 2>$OUT/err_$l.txt; cut -c1-20,180-330 $OUT/voxgo_train_lazy$l.jsonl; tail -3 $OUT/err_$l.txt
done
