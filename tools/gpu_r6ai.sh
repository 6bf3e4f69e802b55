#!/bin/bash
# visit AI: does the training step's clock depend on how many streams the process created before?  (bench.py's S3 secondaries read 7.6 / 3.2 ms
# instead of 5.8 / 2.2 since the frame benches take three streams each)
OUT=gpurun_out/r6ai; mkdir -p $OUT; : > $OUT/burn.jsonl
for n in 0 1 2 3 4 5 6 7 8 16 31 32 33 64 255 270; do
  timeout 300 python tools/bench_train_step.py --steps 24 --blocks 3 --warmup 4 --first-step 10001 --sync-free 1 --lazy-loss 1 --burn-streams $n 2>/dev/null | grep '^{' | sed "s/^{/{\"burn\": $n, /" >> $OUT/burn.jsonl
done
python - <<'PY' | tee $OUT/summary.txt
import json
for l in open("gpurun_out/r6ai/burn.jsonl"):
    d = json.loads(l); print("streams taken before", d["burn"], " sync-free masked S3 step %.3f ms" % d["ms_per_step"], d.get("block_ms"))
PY
