#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_dvgo.py tests/test_dcvgo.py tests/test_checkpoint.py -m gpu -x -q -p no:warnings 2>&1 | tail -8
