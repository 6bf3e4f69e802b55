#!/bin/bash
# visit Z: the frame loop with two views in flight (tests), before the final collection
OUT=gpurun_out/r6z; mkdir -p $OUT
timeout 900 python -m pytest tests/test_checkpoint.py tests/test_dvgo.py -x -q -m gpu -p no:warnings 2>&1 | tail -5 | tee $OUT/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_s1_scale.py -x -q -m gpu -p no:warnings -k "frame_loop or graph or truck" --durations=5 2>&1 | tail -12 | tee $OUT/pytest_b.log
