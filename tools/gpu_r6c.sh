#!/bin/bash
# round 6, visit C: the shade kernel's ablation arms (timing only; UG_ABL_* in csrc/ugrid_shade_pc.h) on S1 (F = 3) and the truck shape (F = 4),
# then the whole GPU suite + the driver's bench command on the shipped library
OUT=gpurun_out/r6c; mkdir -p $OUT
LIBS="build/ab/abl_base.so build/ab/abl_noemb.so build/ab/abl_nol3.so build/ab/abl_noemb_nol3.so build/ab/abl_noconsume.so build/ab/abl_nogather.so"
AB_REPS=2 BENCH_FLAGS="--no-truck --no-proxy" bash tools/gpu_ab.sh r6c $LIBS
mv $OUT/ab.txt $OUT/ab_s1.txt
AB_REPS=1 AB_STEPS=8 BENCH_FLAGS="--no-truck --no-proxy --scene s1b --freq 4 --stepsize 0.5" bash tools/gpu_ab.sh r6c $LIBS
mv $OUT/ab.txt $OUT/ab_truck.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.log
wc -c $OUT/bench_line.json
