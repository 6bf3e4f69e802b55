#!/bin/bash
# Copy the judged artefacts of a tools/gpu_final_r6.sh visit from gpurun_out/<tag> into profiles/r06/.   usage: tools/collect_profiles_r6.sh r6final
set -e
cd "$(dirname "$0")/.."
T=gpurun_out/$1
P=profiles/r06
mkdir -p $P/pmc
rm -f $P/pmc/*.csv
cp $T/pmc_csv/*.csv $P/pmc/
cp $T/pmc_summary.json $P/pmc_summary.json
cp $T/tv_adam_dense_pmc.json $P/tv_adam_dense_pmc.json
cp $T/s1_kernel_stats.csv $P/bench_s1_kernel_stats.csv
cp $T/truck_kernel_stats.csv $P/bench_truck_kernel_stats.csv
tail -1 $T/bench_line.json > $P/bench_s1_line.json
cp $T/bench_detail.json $P/bench_s1_detail.json
[ -s $T/s1_two_frames_in_flight_kernel_stats.csv ] && cp $T/s1_two_frames_in_flight_kernel_stats.csv $P/bench_s1_two_frames_in_flight_kernel_stats.csv
for f in bench_2rank_shared_gpu.json bench_8rank_shared_gpu.json train_steps.jsonl tv_adam_dense.jsonl pytest_gpu.log smoke.log device_code_sha16.txt; do
  [ -s $T/$f ] && cp $T/$f $P/$f
done
python - <<'PY'
import json
d = json.load(open('profiles/r06/pmc_summary.json'))
print("device code", d.get("device_code_sha16"))
for k in ('render_march', 'render_shade'):
    c = d[k]
    print(k, {x: (round(c[x], 4) if isinstance(c.get(x), float) else c.get(x)) for x in ("hbm_bytes", "l2_hit_rate", "l1_hit_rate", "ta_busy_frac", "ta_clocks_per_wave_instruction",
                                                       "lds_array_busy_frac", "rocprofv3_avg_ms")})
b = json.load(open('profiles/r06/bench_s1_line.json'))
print("line: %.3f ms, %.0f Msamples/s, kernels %s, device code %s" % (b['ms_per_step'], b['value'], b['kernels'], b['device_code_sha16']))
print("roofline", b['roofline']); print("roofline_hbm", b.get('roofline_hbm')); print("secondary", b.get('secondary_ms')); print("proxy", b.get('scaling_proxy_N8'))
PY
