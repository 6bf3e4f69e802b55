#!/bin/bash
# round 6, visit J: A/B of the march with wave-uniform records fetched by ONE scalar load (-DUG_MARCH_SCALAR_UNIFORM) against the shipped kernel
LIBS="build/ab/base6.so build/ab/march_su.so"
AB_NAME=s1 AB_REPS=2 BENCH_FLAGS="--no-truck --no-proxy" bash tools/gpu_ab.sh r6j $LIBS
AB_NAME=s1b AB_REPS=1 BENCH_FLAGS="--no-truck --no-proxy --scene s1b" bash tools/gpu_ab.sh r6j $LIBS
AB_NAME=truck AB_REPS=1 AB_STEPS=8 BENCH_FLAGS="--no-truck --no-proxy --scene s1b --freq 4 --stepsize 0.5" bash tools/gpu_ab.sh r6j $LIBS
