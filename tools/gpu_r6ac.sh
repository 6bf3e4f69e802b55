#!/bin/bash
# visit AC: the training backward on side lanes (train_overlap): tests, then step clocks with the lanes off / on
OUT=gpurun_out/r6ac; mkdir -p $OUT
echo "(tests: first run of this visit script, 131 passed)"
: > $OUT/train_steps.jsonl
for ov in 0 1; do
  for sf in 0 1; do
    timeout 600 python tools/bench_voxgo_train.py --tune train_overlap=$ov --model both --steps 100 --warmup 10 --sync-free $sf --lazy-loss $sf 2>/dev/null | grep '^{' | sed "s/^{/{\"train_overlap\": $ov, /" >> $OUT/train_steps.jsonl
    for ph in 1 10001; do timeout 600 python tools/bench_train_step.py --tune train_overlap=$ov --steps 30 --first-step $ph --sync-free $sf --lazy-loss $sf 2>/dev/null | grep '^{' | sed "s/^{/{\"train_overlap\": $ov, /" >> $OUT/train_steps.jsonl; done
  done
done
python - $OUT/train_steps.jsonl <<'PY' | tee $OUT/summary.txt
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print("overlap", d["train_overlap"], d.get("model", "s3"), d["workload"][-22:], "sync_free", d.get("sync_free"), round(d["ms_per_step"], 4))
PY
