#!/bin/bash
# visit AB: more than two frames in flight?
OUT=gpurun_out/r6ab; mkdir -p $OUT
F="--no-cpu-baseline --no-secondary --no-truck --no-proxy --steps 24 --warmup 6"
for n in 2 3 4; do
  timeout 300 python bench.py $F --frames-in-flight $n 2>$OUT/err_s1_$n.log | tail -1 > $OUT/s1_n$n.json
  timeout 400 python bench.py $F --frames-in-flight $n --scene s1b --freq 4 --stepsize 0.5 2>$OUT/err_truck_$n.log | tail -1 > $OUT/truck_n$n.json
  timeout 400 python bench.py $F --frames-in-flight $n --scene s1b 2>$OUT/err_s1b_$n.log | tail -1 > $OUT/s1b_n$n.json
done
python - <<'PY' | tee $OUT/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6ab/*.json")):
    try:
        d = json.load(open(f)); print("%-22s in flight %d  %.3f ms  one stream %.3f ms  frame %s" % (f.split("/")[-1], d["frames_in_flight"], d["ms_per_step"], d["ms_per_step_single_stream"], d.get("frame_sha16")))
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 600 python tools/bench_dvgo.py --steps 20 2>$OUT/dvgo_err.log | tail -1 > $OUT/dvgo_lego_800.json; python -c "
import json; d=json.load(open('$OUT/dvgo_lego_800.json')); print('dvgo', d['ms_per_view'], d['ms_per_view_two_in_flight'], d['ms_n_in_flight'])" | tee -a $OUT/summary.txt
timeout 600 python tools/bench_dcvgo.py --steps 10 2>$OUT/dcvgo_err.log | tail -1 > $OUT/dcvgo_1080p.json; python -c "
import json; d=json.load(open('$OUT/dcvgo_1080p.json')); print('dcvgo', d['ms_per_frame'], d['ms_per_frame_two_in_flight'], d['ms_n_in_flight'])" | tee -a $OUT/summary.txt
