#!/bin/bash
# visit AA: DVGO / DCVGO frame loops with two views in flight (tests + the configs[0] / configs[1] render clocks)
OUT=gpurun_out/r6aa; mkdir -p $OUT
timeout 900 python -m pytest tests/test_checkpoint.py tests/test_dvgo.py tests/test_dcvgo.py -x -q -m gpu -p no:warnings 2>&1 | tail -5 | tee $OUT/pytest_a.log
timeout 600 python tools/bench_dvgo.py --steps 20 2>$OUT/dvgo_err.log | tail -1 > $OUT/dvgo_lego_800.json; cut -c1-700 $OUT/dvgo_lego_800.json
timeout 600 python tools/bench_dcvgo.py --steps 10 2>$OUT/dcvgo_err.log | tail -1 > $OUT/dcvgo_1080p.json; cut -c1-700 $OUT/dcvgo_1080p.json
tail -3 $OUT/dvgo_err.log $OUT/dcvgo_err.log
