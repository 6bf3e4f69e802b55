#!/bin/bash
# visit AG: two vs three frames in flight, alternating repetitions (is the 1-2 % of visit AB real on another board?)
OUT=gpurun_out/r6ag; mkdir -p $OUT
F="--no-cpu-baseline --no-secondary --no-truck --no-proxy --steps 30 --warmup 6"
for rep in 1 2 3; do for n in 2 3; do
  timeout 300 python bench.py $F --frames-in-flight $n 2>$OUT/err.log | tail -1 > $OUT/s1_n${n}_$rep.json
  timeout 400 python bench.py $F --frames-in-flight $n --scene s1b --freq 4 --stepsize 0.5 2>$OUT/err.log | tail -1 > $OUT/truck_n${n}_$rep.json
  timeout 400 python bench.py $F --frames-in-flight $n --scene s1b --stepsize 0.5 2>$OUT/err.log | tail -1 > $OUT/s1b668_n${n}_$rep.json
done; done
python - <<'PY' | tee $OUT/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6ag/*.json")):
    try:
        d = json.load(open(f)); print("%-24s in flight %d  %.3f ms  one stream %.3f ms" % (f.split("/")[-1], d["frames_in_flight"], d["ms_per_step"], d["ms_per_step_single_stream"]))
    except Exception as e:
        print(f, "FAILED", e)
PY
