#!/bin/bash
# Copy the judged artefacts of a tools/gpu_final_r5.sh visit from gpurun_out/<tag> into profiles/r05/.
# usage: tools/collect_profiles_r5.sh r5final
set -e
cd "$(dirname "$0")/.."
T=gpurun_out/$1
P=profiles/r05
mkdir -p $P/pmc
rm -f $P/pmc/*.csv
cp $T/pmc_csv/*.csv $P/pmc/
cp $T/pmc_summary.json $P/pmc_summary.json
cp $T/tv_adam_dense_pmc.json $P/tv_adam_dense_pmc.json
cp $T/s1_kernel_stats.csv $P/bench_s1_kernel_stats.csv
cp $T/truck_kernel_stats.csv $P/bench_truck_kernel_stats.csv
tail -1 $T/bench_line.json > $P/bench_s1_line.json
for f in bench_2rank_shared_gpu.json bench_8rank_shared_gpu.json dcvgo_1080p.json dvgo_lego_800.json voxgo_train.jsonl voxgo_train_lazy_loss.jsonl voxgo_train_op_by_op.jsonl \
         voxgo_train_dvgo_kernel_stats.csv voxgo_train_dcvgo_kernel_stats.csv train_step_s3_masked_kernel_stats.csv voxgo_train_host_profile_dvgo.txt \
         train_step_s3.jsonl pytest_gpu.log smoke.log \
         smi_trace.json smi_trace.csv tv_adam_dense.jsonl rank_share_pmc.jsonl; do
  [ -s $T/$f ] && cp $T/$f $P/$f
done
[ -s $T/ta_lanes.json ] && cp $T/ta_lanes.json $P/microbench_ta_lanes.json
for f in s1_fp64_ground_truth.json train_long_parity.json train_long_parity_dcvgo.json train_long_parity_dvgo.json s1_arbitration_s1.json s1_arbitration_s1b.json truck_f4_frame_parity.json; do
  [ -s gpurun_out/$f ] && cp gpurun_out/$f $P/$f
done
python - <<'PY'
import json
d = json.load(open('profiles/r05/pmc_summary.json'))
print("device code", d.get("device_code_sha16"))
for k in ('render_march', 'render_shade'):
    c = d[k]
    print(k, {x: (round(c[x], 4) if isinstance(c.get(x), float) else c.get(x)) for x in ("hbm_bytes", "l2_hit_rate", "l1_hit_rate", "ta_busy_frac", "ta_clocks_per_wave_instruction",
                                                       "lds_array_busy_frac", "rocprofv3_avg_ms")})
b = json.load(open('profiles/r05/bench_s1_line.json'))
print("line: %.3f ms, %.0f Msamples/s, kernels %s, device code %s" % (b['ms_per_step'], b['value'], {k: round(v['ms'], 3) for k, v in b['kernels'].items()}, b['device_code_sha16']))
r = b['roofline']; print({k: r[k] for k in ('kernel', 'bound', 'frac', 'ta_busy_measured', 'traffic', 'hbm_frac_measured', 'frac_of_hbm_algorithmic', 'pmc_refused')})
print("frame", {k: v for k, v in r['frame'].items() if not k.endswith('note') and k != 'algorithmic_bytes_formula'})
print("roofline_hbm", b.get('roofline_hbm'))
PY
