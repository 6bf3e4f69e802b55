#!/bin/bash
# visit Y: the driver's bench command with two frames in flight by default (the line, its size, the detail file)
OUT=gpurun_out/r6y; mkdir -p $OUT
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.txt < /dev/null
wc -c $OUT/bench_line.json; cp bench_detail.json $OUT/bench_detail.json; tail -3 $OUT/bench_err.txt
cat $OUT/bench_line.json
UGRID_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2> $OUT/bench2_err.txt < /dev/null | grep "^{" | tail -1 > $OUT/bench_2rank_shared_gpu.json
tail -3 $OUT/bench2_err.txt; cut -c1-600 $OUT/bench_2rank_shared_gpu.json
