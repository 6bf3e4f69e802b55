#!/bin/bash
# rolling cell set-up in the producers (4 gather items in flight): 4 + 8 geometry (tune 3) and 6 + 6 (A/B library) against the default
T=${1:-r4v}
mkdir -p gpurun_out/$T
timeout 300 python -m pytest tests/test_gpu_fused.py -q -p no:warnings -m gpu --tb=short -k "geometries" 2>&1 | tail -5 > gpurun_out/$T/pytest.log
tail -2 gpurun_out/$T/pytest.log
run() {   # lib tune label
  UGRID_LIB=$1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --tune shade_pc=$2 2>gpurun_out/$T/err_$3.log | tail -1 > gpurun_out/$T/line_$3.json
  python - $3 gpurun_out/$T/line_$3.json <<'PY' | tee -a gpurun_out/$T/ab.txt
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = {n: round(v["ms"], 3) for n, v in d["kernels"].items()}
    print("%-28s step %.3f ms  %s  frame %s" % (sys.argv[1], d["ms_per_step"], k, d.get("frame_sha16")))
except Exception as e:
    print("%-28s FAILED (%s)" % (sys.argv[1], e))
PY
}
D=$PWD/unboundednerfpytorch_amd/libugrid_hip.so
run $D 2 default_6+6_lds_emb
run $D 5 6+6_global_emb
run $D 2 default_6+6_lds_emb
run $D 5 6+6_global_emb
