#!/bin/bash
# round 5, visit B: the cleaned library (frame hash must equal round 4's), F = 4 in the 12-wave shade geometry (rolling set-up) A/B,
# the microbench with the sample section at equal residency, the fused / capi GPU tests
TAG=${1:-r5b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 120 build/ab/ta_lanes > $OUT/ta_lanes.json 2> $OUT/ta_lanes.err; tail -c 600 $OUT/ta_lanes.json
for i in 1 2; do
  timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2> $OUT/bench_s1_$i.err | tail -1 > $OUT/bench_s1_$i.json
  python - $OUT/bench_s1_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("S1 %.3f ms" % d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["kernels"].items()}, "frame_sha16", d["frame_sha16"], "survivors", d["config"]["survivors_M"])
PY
done
for pc in 2 3 2 3; do
  timeout 300 python bench.py --scene s1b --freq 4 --stepsize 0.5 --no-secondary --no-cpu-baseline --steps 6 --warmup 2 --tune shade_pc=$pc 2> $OUT/bench_truck_pc$pc.err | tail -1 > $OUT/bench_truck_pc$pc.json
  python - $OUT/bench_truck_pc$pc.json $pc <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("truck shade_pc=%s %.3f ms" % (sys.argv[2], d["ms_per_step"]), {k: round(v["ms"], 3) for k, v in d["kernels"].items()}, "frame_sha16", d["frame_sha16"])
PY
done
UGRID_TUNE=shade_pc=3 timeout 600 python -m pytest tests/test_gpu_s1_scale.py -q -p no:warnings -s -k "truck" 2>&1 | tail -8 | tee $OUT/pytest_truck_pc3.log
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_capi.py tests/test_dvgo.py tests/test_dcvgo.py -q -p no:warnings -m gpu 2>&1 | tail -6 | tee $OUT/pytest_fused.log
ls $OUT
