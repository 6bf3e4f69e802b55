#!/bin/bash
# round 6, visit D: the shade kernel's ablation arms on S1 (F = 3) -- visit C's S1 lines were overwritten by its truck run
OUT=gpurun_out/r6d; mkdir -p $OUT
LIBS="build/ab/abl_base.so build/ab/abl_noemb.so build/ab/abl_nol3.so build/ab/abl_noemb_nol3.so build/ab/abl_noconsume.so build/ab/abl_nogather.so"
AB_NAME=s1 AB_REPS=2 BENCH_FLAGS="--no-truck --no-proxy" bash tools/gpu_ab.sh r6d $LIBS
