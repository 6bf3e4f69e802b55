#!/bin/bash
# visit AL: frames-in-flight streams chosen by a concurrency probe (fourier_render.concurrent_streams) vs the first pool streams, after N streams were taken
OUT=gpurun_out/r6al; mkdir -p $OUT; : > $OUT/dvgo.jsonl
for probe in 1 0; do for n in 0 1 2 3 5 30 31; do
  UGRID_STREAM_PROBE=$probe timeout 300 python tools/bench_dvgo.py --steps 20 --burn-streams $n 2>$OUT/err.log | tail -1 | sed "s/^{/{\"probe\": $probe, \"burn\": $n, /" >> $OUT/dvgo.jsonl
done; done
python - <<'PY' | tee $OUT/summary.txt
import json
for l in open("gpurun_out/r6al/dvgo.jsonl"):
    d = json.loads(l); print("probe", d["probe"], "streams taken before", d["burn"], " DVGO lego view: one stream %.3f  two %.3f  three %.3f  four %.3f ms" % (d["ms_per_view"], d["ms_per_view_two_in_flight"], d["ms_n_in_flight"]["3"], d["ms_n_in_flight"]["4"]))
PY
F="--no-cpu-baseline --no-secondary --no-truck --no-proxy --steps 24 --warmup 6"
for probe in 1 0; do UGRID_STREAM_PROBE=$probe timeout 300 python bench.py $F 2>>$OUT/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench S1 probe $probe: %.3f ms in flight, %.3f one stream' % (d['ms_per_step'], d['ms_per_step_single_stream']))" | tee -a $OUT/summary.txt; done
timeout 600 python -m pytest tests/test_checkpoint.py tests/test_dvgo.py tests/test_dcvgo.py -x -q -m gpu -p no:warnings 2>&1 | tail -2 | tee -a $OUT/summary.txt
grep -v amdgpu.ids $OUT/err.log | tail -3
