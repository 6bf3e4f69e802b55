#!/bin/bash
# round 6, visit G: the sync-free tests again (visit F stopped at a key the DirectVoxGO dict does not have), the frame-pair proxy
OUT=gpurun_out/r6g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_voxgo_train.py tests/test_gpu_train_scale.py -q -k "sync_free or capturable" 2>&1 | tail -40 | tee $OUT/pytest_sync_free.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-truck > $OUT/bench_proxy_line.json 2> $OUT/bench_proxy_err.log
cp bench_detail.json $OUT/bench_proxy_detail.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6g/bench_proxy_detail.json"))
px = d.get("scaling_proxy", {})
for N in ("N=2", "N=4", "N=8"):
    for deal, r in px.get(N, {}).items():
        print(N, deal, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if "slowest" in k or "speedup" in k})
PY
