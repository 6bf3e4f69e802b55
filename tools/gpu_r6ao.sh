#!/bin/bash
# visit AO: the DCVGO masked step's clock is the noisiest of the set (0.62-1.04 ms over the round): repetitions with / without the side-stream probe
OUT=gpurun_out/r6ao; mkdir -p $OUT; : > $OUT/dcvgo.jsonl
for rep in 1 2 3; do for pick in 1 0; do
  UGRID_SIDE_STREAM_PICK=$pick timeout 300 python tools/bench_voxgo_train.py --model dcvgo --phase masked --steps 100 --blocks 4 --warmup 10 --sync-free 1 --lazy-loss 1 2>$OUT/err.log | grep '^{' | sed "s/^{/{\"pick\": $pick, /" >> $OUT/dcvgo.jsonl
done; done
python - <<'PY' | tee $OUT/summary.txt
import json
for l in open("gpurun_out/r6ao/dcvgo.jsonl"):
    d = json.loads(l); print("pick", d["pick"], d["workload"][-12:], "%.4f ms" % d["ms_per_step"], d.get("block_ms"))
PY
