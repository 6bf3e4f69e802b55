#!/bin/bash
TAG=${1:-r5o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/prof_voxgo_step_cpu.py dvgo 2>&1 | grep -v amdgpu.ids | head -75 | cut -c1-200 | tee $OUT/voxgo_train_host_profile_dvgo.txt
