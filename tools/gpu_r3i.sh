#!/bin/bash
# round 3 visit i: fused bounded-DVGO march
mkdir -p gpurun_out/r3i
timeout 600 python tools/bench_dvgo.py --out gpurun_out/r3i/dvgo_lego_800.json 2>&1 | tail -5
