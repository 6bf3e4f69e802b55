#!/bin/bash
# visit AQ: the process-wide frame streams (fourier_render.frame_streams: created once, first used back to back): DVGO / DCVGO views from several
# starting points, the S1 frame, the frame-loop tests
OUT=gpurun_out/r6aq; mkdir -p $OUT
for b in 0 2 13 30 77; do timeout 300 python tools/bench_dvgo.py --steps 20 --burn-streams $b 2>>$OUT/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('burn $b  DVGO view: one stream %.3f  two %.3f  three %.3f  four %.3f ms' % (d['ms_per_view'], d['ms_per_view_two_in_flight'], d['ms_n_in_flight']['3'], d['ms_n_in_flight']['4']))" | tee -a $OUT/summary.txt; done
for b in 0 5 17 40; do timeout 400 python tools/bench_dcvgo.py --steps 10 --burn-streams $b 2>>$OUT/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('burn $b  DCVGO frame: one stream %.3f  two %.3f  three %.3f  four %.3f ms' % (d['ms_per_frame'], d['ms_per_frame_two_in_flight'], d['ms_n_in_flight']['3'], d['ms_n_in_flight']['4']))" | tee -a $OUT/summary.txt; done
F="--no-cpu-baseline --no-secondary --no-truck --no-proxy --steps 24 --warmup 6"
for n in 3 3; do timeout 300 python bench.py $F --frames-in-flight $n 2>>$OUT/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in flight $n  S1 %.3f ms  (one stream %.3f)' % (d['ms_per_step'], d['ms_per_step_single_stream']))" | tee -a $OUT/summary.txt; done
timeout 900 python -m pytest tests/test_checkpoint.py tests/test_dvgo.py tests/test_dcvgo.py tests/test_gpu_s1_scale.py -x -q -m gpu -p no:warnings -k "frame_loop or render_view or dvgo or dcvgo or checkpoint" 2>&1 | tail -2 | tee -a $OUT/summary.txt
