#!/bin/bash
# round-4 visit D: lane order inside the 8 x 8 pixel block -- row-major vs Z-order (UGRID_TILE_MORTON) -- times and TA / TCP counters.
cd $GRAFT_REPO_ROOT
T=r4d
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
for rep in 1 2; do
for m in 0 1; do
  UGRID_TILE_MORTON=$m timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$T/line_m${m}_$rep.json
  python - $m $rep <<'PY' | tee -a gpurun_out/r4d/ab.txt
import json, sys
d = json.loads(open("gpurun_out/r4d/line_m%s_%s.json" % (sys.argv[1], sys.argv[2])).read())
g = d.get("secondary_garden_single_sampling", {})
print("morton=%s rep %s  S1 step %.3f %s frame %s | S1b %.3f %s | S=668 %.3f %s" % (
    sys.argv[1], sys.argv[2], d["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["kernels"].items()}, d["frame_sha16"],
    d["secondary"]["ms_per_step"], {k: round(v["ms"], 3) for k, v in d["secondary"]["kernels"].items()},
    g.get("ms_per_step", 0), {k: round(v["ms"], 3) for k, v in g.get("kernels", {}).items()}))
PY
done
done
for m in 0 1; do
  UGRID_TILE_MORTON=$m tools/gpu_pmc.sh $T/pmc_m$m \
    "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" > gpurun_out/$T/pmc_m$m.txt 2>&1
done
cat gpurun_out/$T/ab.txt
grep -h "TA_TA_BUSY\|TOTAL_CACHE_ACC\|TCC_READ_REQ_sum" gpurun_out/$T/pmc_m0.txt gpurun_out/$T/pmc_m1.txt
