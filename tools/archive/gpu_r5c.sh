#!/bin/bash
# round 5, visit C: scaling proxy (every rank's share of N = 2, 4, 8 deals alone on the device) + per-share HBM bytes,
# the multi-rank tests at world 2 / 3 / 8 over gloo on this GPU, the geometry test at F = 3 and 4
TAG=${1:-r5c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2> $OUT/bench_proxy.err | tail -1 > $OUT/bench_proxy.json
python - $OUT/bench_proxy.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("S1 %.3f ms" % d["ms_per_step"], d["frame_sha16"])
p = d.get("scaling_proxy", {})
for N in ("N=2", "N=4", "N=8"):
    for deal, r in p.get(N, {}).items():
        print(N, "%-24s slowest %.3f mean %.3f eff %.3f shares %s" % (deal, r["slowest_share_ms"], r["mean_share_ms"], r["predicted_efficiency"], r["share_ms"]))
if "error" in p: print(p)
PY
bash tools/gpu_rank_share.sh $TAG 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_fused.py -q -p no:warnings -m gpu -k "gloo_ranks or geometries or armed" --durations=12 2>&1 | tail -25 | tee $OUT/pytest_multi.log
ls $OUT
