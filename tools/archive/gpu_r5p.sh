#!/bin/bash
# Round-5 visit P: the one-launch rgbnet input rows (ops.rgbnet_features) -- its test, the models' tests, the training-step legs
OUT=gpurun_out/r5p; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:warnings -k "rgbnet" 2>&1 | tail -5
timeout 1200 python -m pytest tests -m gpu -q -p no:warnings -x -k "voxgo or train or model or golden or dvgo or dcvgo" 2>&1 | tail -5
timeout 600 python tools/bench_voxgo_train.py --model both > $OUT/voxgo_train.jsonl 2>/dev/null; cut -c1-300 $OUT/voxgo_train.jsonl
for ph in 1 10001; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph 2>/dev/null | tail -1 | cut -c1-400
done
