#!/bin/bash
# round 6, visit T: shade A/B -- which consumer drains which producer's ring (like-with-like by SIMD population)
AB_NAME=s1 AB_REPS=3 BENCH_FLAGS="--no-truck --no-proxy" bash tools/gpu_ab.sh r6t build/ab/base7.so build/ab/pair_like.so
AB_NAME=truck AB_REPS=2 AB_STEPS=8 BENCH_FLAGS="--no-truck --no-proxy --scene s1b --freq 4 --stepsize 0.5" bash tools/gpu_ab.sh r6t build/ab/base7.so build/ab/pair_like.so
