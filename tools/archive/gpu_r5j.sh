#!/bin/bash
TAG=${1:-r5j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for pc in 2 3 4 5; do
  UGRID_LIB=$GRAFT_REPO_ROOT/build/ab/shade_exp.so timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-proxy --steps 10 --warmup 3 --tune shade_pc=$pc 2> $OUT/err_$pc.txt | tail -1 > $OUT/b_$pc.json
  python - $OUT/b_$pc.json $pc <<'PY' | tee -a $OUT/shade_roll_ab.txt
import json, sys
d = json.load(open(sys.argv[1])); print("shade_pc=%s %.3f ms" % (sys.argv[2], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["kernels"].items()}, "frame_sha16", d["frame_sha16"])
PY
done
done
timeout 600 python -m pytest tests/test_gpu_multi.py -q -p no:warnings -m gpu -k "data_parallel" -s 2>&1 | tail -6 | cut -c1-600 | tee $OUT/pytest_dp.log
