#!/bin/bash
# 4 + 8 shade geometry: bit-equality test, then S1 A/B (shade_pc = 2 vs 3) on the same library
T=${1:-r4m}
mkdir -p gpurun_out/$T
timeout 300 python -m pytest tests/test_gpu_fused.py -q -p no:warnings -m gpu --tb=short -k "geometries" 2>&1 | tail -15 > gpurun_out/$T/pytest.log
tail -3 gpurun_out/$T/pytest.log
for pc in 2 3 2 3; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --tune shade_pc=$pc 2>gpurun_out/$T/err_$pc.log | tail -1 > gpurun_out/$T/line_$pc.json
  python - $pc gpurun_out/$T/line_$pc.json <<'PY' | tee -a gpurun_out/$T/ab.txt
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = {n: round(v["ms"], 3) for n, v in d["kernels"].items()}
    print("shade_pc=%s  step %.3f ms  %s  frame %s" % (sys.argv[1], d["ms_per_step"], k, d.get("frame_sha16")))
except Exception as e:
    print("shade_pc=%s FAILED (%s)" % (sys.argv[1], e))
PY
done
