#!/bin/bash
# One visit collecting the judged artefacts of round 3 on the FINAL binary (every step under its own timeout).
# usage (through gpurun): tools/gpu_final_r3.sh <tag>   -> gpurun_out/<tag>/..., then tools/collect_profiles_r3.sh <tag> here
TAG=${1:-r3final}
OUT=gpurun_out/$TAG
mkdir -p $OUT/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import bench; print(bench.device_code_sha16()); print(bench.lib_sha16())" 2>/dev/null | tail -2 > $OUT/device_code_sha16.txt
cat $OUT/device_code_sha16.txt
# 1. PMC passes, one counter group per rocprofv3 run (FETCH_SIZE and WRITE_SIZE together abort the tool)
bash tools/gpu_pmc.sh $TAG "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" > $OUT/pmc_log.txt 2>&1
tail -12 $OUT/pmc_log.txt
# 2. kernel trace + stats of the bench command
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/$OUT/prof_bench.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "s1_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/prof/s1_kernel_stats.csv.keep
rm -rf $(find $OUT/prof -mindepth 1 -maxdepth 1 -type d) 2>/dev/null; [ -f $OUT/prof/s1_kernel_stats.csv.keep ] && mv $OUT/prof/s1_kernel_stats.csv.keep $OUT/prof/s1_kernel_stats.csv
head -5 $OUT/prof/s1_kernel_stats.csv | cut -c1-200
# 3. summarise the counters NOW so that the bench line of step 4 can merge them (same box, same binary)
mkdir -p profiles/r03/pmc; rm -f profiles/r03/pmc/*.csv; i=0
for d in $OUT/pmc_*/; do i=$((i+1)); f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" profiles/r03/pmc/pmc_pass_$i.csv; done
python tools/pmc_summarize.py profiles/r03/pmc $(head -1 $OUT/device_code_sha16.txt) > $OUT/pmc_summary_print.txt 2>&1; cp profiles/r03/pmc_summary.json $OUT/pmc_summary.json
mkdir -p $OUT/pmc_csv; cp profiles/r03/pmc/*.csv $OUT/pmc_csv/ 2>/dev/null
# 4. the bench line itself (CPU baseline = reference Python, arbitration, S1b, S3), then the self-launched 2-rank run
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt < /dev/null; tail -c 600 $OUT/bench_line.json; echo
UGRID_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-chunks 4 2> $OUT/bench2_err.txt < /dev/null | grep "^{" | tail -1 > $OUT/bench_2rank_shared_gpu.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null < /dev/null | grep "^{" | tail -1 > $OUT/bench_dist_1rank_rccl.json
# 5. fused DCVGO at 1080p, shade phase profile of the instrumented build, shuffled-ray frame through the ray-order guard
timeout 900 python tools/bench_dcvgo.py --out $OUT/dcvgo_1080p.json 2>$OUT/dcvgo_err.txt | cut -c1-600; tail -2 $OUT/dcvgo_err.txt | grep -v amdgpu.ids
UGRID_LIB=build/ab/lib_pc12_prof.so timeout 300 python tools/gpu_shade_pc_prof.py s1 2 2>/dev/null > $OUT/shade_pc12_phases.txt; head -8 $OUT/shade_pc12_phases.txt
timeout 300 python tools/gpu_ray_order.py > $OUT/ray_order_guard.json 2>/dev/null; cat $OUT/ray_order_guard.json | cut -c1-500
# 5b. bounded DVGO at the lego size; S3 training step, both TV phases (+ without the per-step host read of loss / psnr); kernel
#     timeline of one masked-phase step
timeout 600 python tools/bench_dvgo.py --out $OUT/dvgo_lego_800.json 2>/dev/null | cut -c1-400
: > $OUT/train_step_s3.jsonl
for ph in 1 10001; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph 2>/dev/null | tail -1 >> $OUT/train_step_s3.jsonl
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph --lazy-loss 1 2>/dev/null | tail -1 >> $OUT/train_step_s3.jsonl
done
python - $OUT/train_step_s3.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['tv_phase'], '%.3f ms' % d['ms_per_step'], {k: round(v, 3) for k, v in d['phases_ms'].items()}, d['survivors_M'], d.get('k0_grad_lines_touched_frac'))
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ts -o p -- python $R/tools/bench_train_step.py --steps 20 --first-step 10001 > /tmp/log_ts.txt 2>&1 < /dev/null )
f=$(find /tmp/prof_ts -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/step_timeline.py "$f" --step 15 > $OUT/train_step_timeline_masked_final.txt; tail -1 $OUT/train_step_timeline_masked_final.txt
# 6. smoke + the whole -m gpu suite
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 2400 python -m pytest tests -m gpu -q -p no:warnings 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
ls -la $OUT
