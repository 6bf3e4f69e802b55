#!/bin/bash
# One visit collecting the judged artefacts of round 4 on the FINAL binary (every step under its own timeout).
# usage (through gpurun): tools/gpu_final_r4.sh <tag>   -> gpurun_out/<tag>/..., then tools/collect_profiles_r4.sh <tag> here
TAG=${1:-r4final}
OUT=gpurun_out/$TAG
mkdir -p $OUT/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -c "import bench; print(bench.device_code_sha16()); print(bench.lib_sha16())" 2>/dev/null | tail -2 > $OUT/device_code_sha16.txt
cat $OUT/device_code_sha16.txt
# 1. PMC passes, one counter group per rocprofv3 run (groups: tools/pmc_summarize.py PMC_GROUPS + the request-size counters)
bash tools/gpu_pmc.sh $TAG "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAVES" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" > $OUT/pmc_log.txt 2>&1
tail -6 $OUT/pmc_log.txt
# 2. kernel trace + stats of the bench command
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $R/$OUT/prof_bench.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "s1_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/s1_kernel_stats.csv
rm -rf $OUT/prof
head -5 $OUT/s1_kernel_stats.csv | cut -c1-200
# 3. summarise the counters NOW so that the bench line of step 4 can merge them (same box, same binary); FETCH_SIZE calibration first
mkdir -p $OUT/fetch
i=0
for ctrs in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/fc_$i -o p -- $R/build/ab/fetch_calib > $R/$OUT/fetch/fetch_calib_$i.out 2>&1 )
  f=$(find /tmp/fc_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/fetch/fetch_pass_$i.csv
done
grep requested_bytes $OUT/fetch/fetch_calib_1.out > $OUT/fetch/fetch_calib.json
mkdir -p profiles/r04/pmc; rm -f profiles/r04/pmc/*.csv; cp $OUT/pmc_csv/*.csv profiles/r04/pmc/
python tools/microbench/fetch_calib_report.py $OUT/fetch profiles/r04/microbench_fetch_calib.json > $OUT/fetch/report.txt 2>&1
python tools/pmc_summarize.py profiles/r04/pmc $(head -1 $OUT/device_code_sha16.txt) profiles/r04/pmc_summary.json > $OUT/pmc_summary_print.txt 2>&1
cp profiles/r04/pmc_summary.json $OUT/pmc_summary.json; cp profiles/r04/microbench_fetch_calib.json $OUT/
# 4. the bench line itself (CPU baseline = reference Python, arbitration + fp64 ground truth, S1b, S = 668, S3), the self-launched 2-rank run
timeout 1200 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt < /dev/null; tail -c 400 $OUT/bench_line.json; echo
UGRID_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-chunks 4 2> $OUT/bench2_err.txt < /dev/null | grep "^{" | tail -1 > $OUT/bench_2rank_shared_gpu.json
# 5. power / clock trace of the two kernels; fused DCVGO at 1080p; bounded DVGO at the lego size
timeout 300 python tools/smi_trace.py $OUT/smi_trace -- python tools/smi_phases.py > $OUT/smi_trace_stdout.txt 2>&1; tail -30 $OUT/smi_trace_stdout.txt | head -40
timeout 900 python tools/bench_dcvgo.py --out $OUT/dcvgo_1080p.json 2>$OUT/dcvgo_err.txt | cut -c1-400
timeout 600 python tools/bench_dvgo.py --out $OUT/dvgo_lego_800.json 2>/dev/null | cut -c1-400
: > $OUT/train_step_s3.jsonl
for ph in 1 10001; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph 2>/dev/null | tail -1 >> $OUT/train_step_s3.jsonl
done
python - $OUT/train_step_s3.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['tv_phase'], '%.3f ms' % d['ms_per_step'], {k: round(v, 3) for k, v in d['phases_ms'].items()}, d['survivors_M'], d.get('k0_grad_lines_touched_frac'), d.get('roofline_tv_adam_dense'))
PY
timeout 600 python tools/bench_voxgo_train.py --model both > $OUT/voxgo_train.jsonl 2>/dev/null; cut -c1-300 $OUT/voxgo_train.jsonl
timeout 300 python tools/bench_voxgo_train.py --model both --fused 0 --steps 10 > $OUT/voxgo_train_composed.jsonl 2>/dev/null
# 6. smoke + the whole -m gpu suite
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 2700 python -m pytest tests -m gpu -q -p no:warnings 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
ls $OUT
