#!/bin/bash
# visit AK: the side stream chosen by a one-off overlap probe (sharded_adam._low_priority_stream): the step's clock for the stream counts that were slow
OUT=gpurun_out/r6ak; mkdir -p $OUT; : > $OUT/burn.jsonl
for pick in 1 0; do for n in 0 6 32 64 255 5; do
  UGRID_SIDE_STREAM_PICK=$pick timeout 300 python tools/bench_train_step.py --steps 24 --blocks 3 --warmup 4 --first-step 10001 --sync-free 1 --lazy-loss 1 --burn-streams $n 2>$OUT/err.log | grep '^{' | sed "s/^{/{\"pick\": $pick, \"burn\": $n, /" >> $OUT/burn.jsonl
done; done
python - <<'PY' | tee $OUT/summary.txt
import json
for l in open("gpurun_out/r6ak/burn.jsonl"):
    d = json.loads(l); print("pick", d["pick"], "streams taken before", d["burn"], " sync-free masked S3 step %.3f ms" % d["ms_per_step"], " probe (side, main ms):", d.get("side_stream_pick"))
PY
tail -3 $OUT/err.log
