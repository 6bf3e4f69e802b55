#!/bin/bash
TAG=${1:-r5e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in poll device stream; do
  timeout 300 python tools/diag_share_stall.py $m 2>&1 | grep -v "amdgpu.ids" | tail -17 | tee -a $OUT/diag_share_stall.txt
done
