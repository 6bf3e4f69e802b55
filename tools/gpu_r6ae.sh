#!/bin/bash
# visit AE: do the run-time geometry knobs (march waves per SIMD, 8- vs 12-wave shade) have another optimum with two frames in flight?
OUT=gpurun_out/r6ae; mkdir -p $OUT
F="--no-cpu-baseline --no-secondary --no-truck --no-proxy --steps 24 --warmup 6"
for mw in 6 5 4; do for pc in 2 1; do
  timeout 300 python bench.py $F --tune march_waves=$mw --tune shade_pc=$pc 2>$OUT/err.log | tail -1 > $OUT/s1_mw${mw}_pc${pc}.json
  timeout 400 python bench.py $F --tune march_waves=$mw --tune shade_pc=$pc --scene s1b --freq 4 --stepsize 0.5 2>$OUT/err.log | tail -1 > $OUT/truck_mw${mw}_pc${pc}.json
done; done
python - <<'PY' | tee $OUT/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6ae/*.json")):
    try:
        d = json.load(open(f)); print("%-24s two in flight %.3f ms  one stream %.3f ms  %s  frame %s" % (f.split("/")[-1], d["ms_per_step"], d["ms_per_step_single_stream"], d["kernels"], d.get("frame_sha16")))
    except Exception as e:
        print(f, "FAILED", e)
PY
