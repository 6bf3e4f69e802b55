#!/bin/bash
TAG=${1:-r5n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_multi.py -q -p no:warnings -m gpu -k "data_parallel" -s 2>&1 | tail -8 | cut -c1-900 | tee $OUT/pytest_dp.log
