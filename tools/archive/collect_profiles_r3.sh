#!/bin/bash
# Copy the judged artefacts of a tools/gpu_final_r3.sh visit from gpurun_out/<tag> into profiles/r03/.
# usage: tools/collect_profiles_r3.sh r3final
set -e
cd "$(dirname "$0")/.."
T=gpurun_out/$1
P=profiles/r03
mkdir -p $P/pmc
rm -f $P/pmc/*.csv
cp $T/pmc_csv/*.csv $P/pmc/
cp $T/pmc_summary.json $P/pmc_summary.json
cp $T/prof/s1_kernel_stats.csv $P/bench_s1_kernel_stats.csv
tail -1 $T/bench_line.json > $P/bench_s1_line.json
for f in bench_2rank_shared_gpu.json bench_dist_1rank_rccl.json dcvgo_1080p.json dvgo_lego_800.json train_step_s3.jsonl train_step_timeline_masked_final.txt shade_pc12_phases.txt ray_order_guard.json pytest_gpu.log smoke.log; do
  [ -s $T/$f ] && cp $T/$f $P/$f
done
python - <<'PY'
import json
d = json.load(open('profiles/r03/pmc_summary.json'))
print("device code", d.get("device_code_sha16"))
for k in ('render_march', 'render_shade'):
    c = d[k]; cyc = c['gui_active_cycles']; alg = {'render_march': 118974873600, 'render_shade': 70113144960}[k]
    print(k, "hbm GB %.2f" % (c['hbm_bytes'] / 1e9), "valu busy %.3f" % (c['valu_insts'] * 2 / 1024 / cyc), "mfma busy %.3f" % (c['mfma_busy_cycles'] / 1024 / cyc),
          "l1 busy %.3f" % (alg / (256 * 64) / cyc), "L2 hit %.2f" % c['l2_hit_rate'], "cycles %.3g" % cyc)
b = json.load(open('profiles/r03/bench_s1_line.json'))
print("line: %.3f ms, %.0f Msamples/s, kernels %s, device code %s" % (b['ms_per_step'], b['value'], {k: round(v['ms'], 3) for k, v in b['kernels'].items()}, b['device_code_sha16']))
r = b['roofline']; print({k: r[k] for k in ('kernel', 'bound', 'frac', 'traffic', 'pmc_refused')})
PY
head -4 $P/bench_s1_kernel_stats.csv | cut -c1-160
