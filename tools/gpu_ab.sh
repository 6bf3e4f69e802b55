#!/bin/bash
# A/B of alternative builds of libugrid_hip.so on the S1 frame (one gpurun visit).
#   usage: tools/gpu_ab.sh <tag> <lib.so> [<lib.so> ...]      -> gpurun_out/<tag>/ab.txt
# Each library runs `bench.py` in its own process (UGRID_LIB); the table shows the live HIP-event kernel times, the step time
# and the frame hash (same hash = bit-identical frame).  Libraries are built here, in the CPU container, e.g.
#   UG_OUT=$PWD/build/ab/dsadd.so UG_SHADE_FLAGS=-DUG_ACC_DSADD bash unboundednerfpytorch_amd/csrc/build.sh
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
STEPS=${AB_STEPS:-12}
for lib in "$@"; do
  for rep in $(seq 1 ${AB_REPS:-1}); do
    LINE=$OUT/line_${AB_NAME:-ab}_$(basename $lib)_$rep.json
    UGRID_LIB=$PWD/$lib timeout 300 python bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-secondary $BENCH_FLAGS 2>$OUT/err_$(basename $lib).log | tail -1 > $LINE
    python - "$lib" $LINE <<'PY' | tee -a $OUT/ab.txt
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = {n: round(v["ms"] if isinstance(v, dict) else v, 3) for n, v in d["kernels"].items()}      # (round 6: the compact line holds plain numbers)
    print("%-34s step %.3f ms  %s  frame %s  code %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], k, d.get("frame_sha16"), d.get("device_code_sha16")))
except Exception as e:
    print("%-34s FAILED (%s)" % (sys.argv[1], e))
PY
  done
done
