#!/bin/bash
# Copy the judged artefacts of a tools/gpu_pmc.sh + kernel-trace visit from gpurun_out/<tag> into profiles/r02/ and
# rebuild profiles/r02/pmc_summary.json.   usage: tools/collect_profiles.sh r2final
set -e
cd "$(dirname "$0")/.."
T=gpurun_out/$1
mkdir -p profiles/r02/pmc
rm -f profiles/r02/pmc/*.csv
i=0
for d in $T/pmc_*/; do i=$((i+1)); cp $d/p_counter_collection.csv profiles/r02/pmc/pmc_pass_$i.csv; done
cp $T/prof/s1_kernel_stats.csv profiles/r02/bench_s1_kernel_stats.csv
[ -f $T/dropin_ops.txt ] && cp $T/dropin_ops.txt profiles/r02/dropin_ops_vs_reference_kernels.txt
[ -f $T/train_step.json ] && cp $T/train_step.json profiles/r02/train_step_s3.json
[ -s $T/train_step_inorder.json ] && cp $T/train_step_inorder.json profiles/r02/train_step_s3_inorder.json
[ -s $T/train_step_kernels_per_step.txt ] && cp $T/train_step_kernels_per_step.txt profiles/r02/train_step_s3_kernels_per_step.txt
[ -s $T/tv_adam_layouts.json ] && cp $T/tv_adam_layouts.json profiles/r02/tv_adam_layouts.json
[ -s $T/bench_line.json ] && tail -1 $T/bench_line.json > profiles/r02/bench_s1_line.json
[ -s $T/bench_dist_1rank_rccl.json ] && cp $T/bench_dist_1rank_rccl.json profiles/r02/bench_dist_1rank_rccl.json
python tools/pmc_summarize.py profiles/r02/pmc $(tail -1 $T/lib_sha16.txt) > /dev/null
python - <<'PY'
import json
d=json.load(open('profiles/r02/pmc_summary.json'))
print("lib", d["lib_sha16"])
for k in ('render_march','render_shade'):
    c=d[k]; cyc=c['gui_active_cycles']; alg={'render_march':118974873600,'render_shade':70113155712}[k]
    print(k, "hbm GB %.2f" % (c['hbm_bytes']/1e9), "valu busy %.3f" % (c['valu_insts']*2/1024/cyc), "mfma busy %.3f" % (c['mfma_busy_cycles']/1024/cyc), "l1 busy %.3f" % (alg/(256*64)/cyc), "L2 hit %.2f" % c['l2_hit_rate'], "cycles %.3g" % cyc)
PY
head -4 profiles/r02/bench_s1_kernel_stats.csv | cut -c1-160
