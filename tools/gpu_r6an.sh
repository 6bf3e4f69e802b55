#!/bin/bash
# visit AN: with the package's default of 16 hardware queues: S1 frame, DVGO view, the S3 step after 0 / 6 / 32 / 255 streams, the stream tests
OUT=gpurun_out/r6an; mkdir -p $OUT
F="--no-cpu-baseline --no-secondary --no-truck --no-proxy --steps 24 --warmup 6"
for n in 3 4; do timeout 300 python bench.py $F --frames-in-flight $n 2>$OUT/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in flight $n  S1 %.3f ms  (one stream %.3f)' % (d['ms_per_step'], d['ms_per_step_single_stream']))" | tee -a $OUT/summary.txt; done
for b in 0 2 30; do timeout 300 python tools/bench_dvgo.py --steps 20 --burn-streams $b 2>>$OUT/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('burn $b  DVGO view: one stream %.3f  two %.3f  three %.3f  four %.3f ms' % (d['ms_per_view'], d['ms_per_view_two_in_flight'], d['ms_n_in_flight']['3'], d['ms_n_in_flight']['4']))" | tee -a $OUT/summary.txt; done
for n in 0 6 32 255; do timeout 300 python tools/bench_train_step.py --steps 24 --blocks 3 --warmup 4 --first-step 10001 --sync-free 1 --lazy-loss 1 --burn-streams $n 2>>$OUT/err.log | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('burn $n  S3 masked sync-free %.3f ms  probe %s' % (d['ms_per_step'], d.get('side_stream_pick')))" | tee -a $OUT/summary.txt; done
timeout 900 python -m pytest tests/test_gpu_train_scale.py tests/test_checkpoint.py tests/test_dvgo.py tests/test_dcvgo.py -x -q -m gpu -p no:warnings 2>&1 | tail -2 | tee -a $OUT/summary.txt
python -c "import os, unboundednerfpytorch_amd; print('GPU_MAX_HW_QUEUES', os.environ.get('GPU_MAX_HW_QUEUES'))" | tee -a $OUT/summary.txt
