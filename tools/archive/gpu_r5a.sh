#!/bin/bash
# round 5, visit A: TA active-lane microbench, the truck-shaped (F = 4) frame parity + clock, the tightened S1 assertions
TAG=${1:-r5a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 120 build/ab/ta_lanes > $OUT/ta_lanes.json 2> $OUT/ta_lanes.err; tail -c 1500 $OUT/ta_lanes.json
timeout 900 python -m pytest tests/test_gpu_s1_scale.py -q -p no:warnings -s -k "truck or pe8" 2>&1 | tail -25 | tee $OUT/pytest_truck.log
timeout 300 python bench.py --scene s1b --freq 4 --stepsize 0.5 --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > $OUT/bench_truck_headline.json 2> $OUT/bench_truck.err
tail -c 1200 $OUT/bench_truck_headline.json; tail -3 $OUT/bench_truck.err
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o truck -- python $R/bench.py --scene s1b --freq 4 --stepsize 0.5 --no-secondary --no-cpu-baseline --steps 4 --warmup 1 > $R/$OUT/prof_truck.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "truck_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/truck_kernel_stats.csv
rm -rf $OUT/prof
head -6 $OUT/truck_kernel_stats.csv | cut -c1-220
timeout 1200 python -m pytest tests/test_gpu_s1_scale.py -q -p no:warnings -s -k "headline or tail" 2>&1 | tail -25 | tee $OUT/pytest_s1.log
ls $OUT
