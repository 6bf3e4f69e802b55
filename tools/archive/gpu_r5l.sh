#!/bin/bash
TAG=${1:-r5l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.txt
T1=$(date +%s.%N)
echo "bench.py default run wall clock: $(echo "$T1 - $T0" | bc) s" | tee $OUT/bench_wall.txt
python - $OUT/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("S1 %.3f ms" % d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["kernels"].items()}, d["frame_sha16"], "truck %.2f" % d["secondary_truck_render"]["ms_per_step"], "proxy N=8 bands x%.2f" % d["scaling_proxy"]["N=8"]["contiguous_bands"]["predicted_speedup"], "roofline_hbm %.3f" % d["roofline_hbm"]["frac"], "pmc_refused", d["roofline"]["pmc_refused"])
PY
