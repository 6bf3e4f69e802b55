#!/bin/bash
TAG=${1:-r5i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -p no:warnings --durations=20 -x 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
