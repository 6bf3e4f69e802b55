#!/bin/bash
# round 3 visit i: new shade shapes (viewbase_pe 8, rgbnet_dim 15)
mkdir -p gpurun_out/r3i
timeout 1200 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -p no:warnings 2>&1 | tail -15 > gpurun_out/r3i/pytest_fused.log
cat gpurun_out/r3i/pytest_fused.log
