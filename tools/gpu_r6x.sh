#!/bin/bash
# visit X: the frame-pair streams on the WHOLE frame at N = 1 (the proxy only measured 1/2 .. 1/8 shares): S1 and the truck shape
OUT=gpurun_out/r6x; mkdir -p $OUT
F="--no-cpu-baseline --no-secondary --no-truck --no-proxy --steps 20 --warmup 5"
for rep in 1 2; do
  for fp in 0 1; do
    timeout 300 python bench.py $F --frame-pair $fp 2>$OUT/err_s1_$fp.log | tail -1 > $OUT/s1_fp${fp}_$rep.json
    timeout 400 python bench.py $F --frame-pair $fp --scene s1b --freq 4 --stepsize 0.5 2>$OUT/err_truck_$fp.log | tail -1 > $OUT/truck_fp${fp}_$rep.json
  done
done
python - <<'PY' | tee $OUT/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6x/*.json")):
    try:
        d = json.load(open(f)); print("%-28s %.3f ms  %s  frame %s" % (f.split("/")[-1], d["ms_per_step"], d.get("kernels"), d.get("frame_sha16")))
    except Exception as e:
        print(f, "FAILED", e)
PY
