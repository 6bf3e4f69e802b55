#!/bin/bash
# visit AM: hardware queues (GPU_MAX_HW_QUEUES, default 4) x frames in flight on the S1 frame and the DVGO view
OUT=gpurun_out/r6am; mkdir -p $OUT
F="--no-cpu-baseline --no-secondary --no-truck --no-proxy --steps 24 --warmup 6"
for q in 4 8 2 16; do for n in 2 3 4; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py $F --frames-in-flight $n 2>$OUT/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('queues $q  in flight $n  S1 %.3f ms  (one stream %.3f)' % (d['ms_per_step'], d['ms_per_step_single_stream']))" | tee -a $OUT/summary.txt
done
GPU_MAX_HW_QUEUES=$q timeout 300 python tools/bench_dvgo.py --steps 20 2>>$OUT/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('queues $q  DVGO view: one stream %.3f  two %.3f  three %.3f  four %.3f ms' % (d['ms_per_view'], d['ms_per_view_two_in_flight'], d['ms_n_in_flight']['3'], d['ms_n_in_flight']['4']))" | tee -a $OUT/summary.txt
done
