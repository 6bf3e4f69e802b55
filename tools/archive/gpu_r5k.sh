#!/bin/bash
TAG=${1:-r5k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_time.txt
grep "Elapsed (wall clock)\|Maximum resident" $OUT/bench_time.txt
python - $OUT/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("S1 %.3f ms" % d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["kernels"].items()}, d["frame_sha16"], "truck %.2f" % d["secondary_truck_render"]["ms_per_step"], "proxy N=8 bands x%.2f" % d["scaling_proxy"]["N=8"]["contiguous_bands"]["predicted_speedup"], "roofline_hbm %.3f" % d["roofline_hbm"]["frac"], "pmc_refused", d["roofline"]["pmc_refused"])
PY
UGRID_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench2_err.txt | grep "^{" | tail -1 > $OUT/bench2.json
python - $OUT/bench2.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("2 ranks:", d["config"]["parallelism"], d["assembled_frame_equals_single_rank_frame"], [r["rays"] for r in d["per_rank"]])
PY
timeout 3000 python -m pytest tests -m gpu -q -p no:warnings -x 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
