#!/bin/bash
# round 6, visit A: the new one-line bench output under the driver's command + the native-step spread
OUT=gpurun_out/r6a; mkdir -p $OUT
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.log
wc -c $OUT/bench_line.json; tail -c 3000 $OUT/bench_line.json
cp bench_detail.json $OUT/ 2>/dev/null
timeout 1500 python tools/native_step_spread.py --reps 20 --out $OUT/native_step_spread.json 2>&1 | grep -v Warning | tail -8
