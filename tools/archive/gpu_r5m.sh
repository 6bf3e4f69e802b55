#!/bin/bash
TAG=${1:-r5m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for w in 6 7 8; do
  UGRID_LIB=$GRAFT_REPO_ROOT/build/ab/march_w.so timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-proxy --steps 10 --warmup 3 --tune march_waves=$w 2> $OUT/err_$w.txt | tail -1 > $OUT/b_$w.json
  python - $OUT/b_$w.json $w <<'PY' | tee -a $OUT/march_waves_ab.txt
import json, sys
d = json.load(open(sys.argv[1])); print("march_waves=%s %.3f ms" % (sys.argv[2], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["kernels"].items()}, "frame_sha16", d["frame_sha16"])
PY
done
done
