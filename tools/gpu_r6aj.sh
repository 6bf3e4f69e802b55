#!/bin/bash
# visit AJ: the same with more hardware queues (GPU_MAX_HW_QUEUES, default 4)
OUT=gpurun_out/r6aj; mkdir -p $OUT; : > $OUT/burn.jsonl
for q in 8 16 2; do for n in 0 6 32 64 255 5; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/bench_train_step.py --steps 24 --blocks 3 --warmup 4 --first-step 10001 --sync-free 1 --lazy-loss 1 --burn-streams $n 2>/dev/null | grep '^{' | sed "s/^{/{\"queues\": $q, \"burn\": $n, /" >> $OUT/burn.jsonl
done; done
python - <<'PY' | tee $OUT/summary.txt
import json
for l in open("gpurun_out/r6aj/burn.jsonl"):
    d = json.loads(l); print("GPU_MAX_HW_QUEUES", d["queues"], "streams taken before", d["burn"], " sync-free masked S3 step %.3f ms" % d["ms_per_step"])
PY
