#!/bin/bash
# One visit collecting the judged artefacts of round 5 on the FINAL binary (every step under its own timeout).
# usage (through gpurun): tools/gpu_final_r5.sh <tag>   -> gpurun_out/<tag>/..., then tools/collect_profiles_r5.sh <tag> here
TAG=${1:-r5final}
OUT=gpurun_out/$TAG
mkdir -p $OUT/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -c "import bench; print(bench.device_code_sha16()); print(bench.lib_sha16())" 2>/dev/null | tail -2 > $OUT/device_code_sha16.txt
cat $OUT/device_code_sha16.txt
CODE=$(head -1 $OUT/device_code_sha16.txt)
# 1. PMC passes, one counter group per rocprofv3 run (groups: tools/pmc_summarize.py PMC_GROUPS + the request-size counters)
bash tools/gpu_pmc.sh $TAG "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAVES" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" > $OUT/pmc_log.txt 2>&1
tail -4 $OUT/pmc_log.txt
# 2. kernel trace + stats of the bench command, and of the truck-shaped frame (the F = 4 shade geometry's row)
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-proxy > $R/$OUT/prof_bench.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "s1_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/s1_kernel_stats.csv
rm -rf $OUT/prof; mkdir -p $OUT/prof
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o truck -- python $R/bench.py --scene s1b --freq 4 --stepsize 0.5 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-proxy > $R/$OUT/prof_truck.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "truck_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/truck_kernel_stats.csv
rm -rf $OUT/prof
head -4 $OUT/s1_kernel_stats.csv | cut -c1-200; grep "k_shade_pc\|k_march" $OUT/truck_kernel_stats.csv | cut -c1-200
# 3. the dense TV + Adam pass alone: time per mode, HBM bytes by request counters
for t in 3 2; do
  timeout 200 python tools/bench_tv_adam_dense.py --tune tv_xcd=$t 2>/dev/null | tail -1 | tee -a $OUT/tv_adam_dense.jsonl | cut -c1-250
done
i=0
for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/tv_$i -o p -- python $R/tools/bench_tv_adam_dense.py --reps 4 > $R/$OUT/tv_pmc_$i.log 2>&1 )
  f=$(find /tmp/tv_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep "Kernel_Name\|k_tv_cl" "$f" > $OUT/tv_pmc_pass_$i.csv
done
python - $OUT $CODE <<'PY'
import csv, collections, json, sys
out, code = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for i in (1, 2):
    try:
        for r in csv.DictReader(open("%s/tv_pmc_pass_%d.csv" % (out, i))):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e:
        print("tv pmc pass", i, e)
m = {c: sum(v) / len(v) for c, v in agg.items()}
rd = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
wr64 = m.get("TCC_EA0_WRREQ_64B_sum", 0)
wr = 64 * wr64 + 32 * (m.get("TCC_EA0_WRREQ_sum", 0) - wr64)
res = {"_comment": "ugrid_tv_adam_dense_cl (k_tv_cl_slab) on the S3 k0 grid, per-launch means of two rocprofv3 --pmc passes over tools/bench_tv_adam_dense.py",
       "device_code_sha16": code, "counters": m, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
       "algorithmic_bytes": 7 * 9 * 12 * 200 ** 3 * 4}
json.dump(res, open(out + "/tv_adam_dense_pmc.json", "w"), indent=1)
print({k: v for k, v in res.items() if k not in ("counters", "_comment")})
PY
# 4. summarise the counters NOW so that the bench line of step 5 merges them (same box, same binary)
mkdir -p profiles/r05/pmc; rm -f profiles/r05/pmc/*.csv; cp $OUT/pmc_csv/*.csv profiles/r05/pmc/
cp $OUT/tv_adam_dense_pmc.json profiles/r05/tv_adam_dense_pmc.json
cp $OUT/s1_kernel_stats.csv profiles/r05/bench_s1_kernel_stats.csv
python tools/pmc_summarize.py profiles/r05/pmc $CODE profiles/r05/pmc_summary.json $OUT/s1_kernel_stats.csv > $OUT/pmc_summary_print.txt 2>&1
cp profiles/r05/pmc_summary.json $OUT/pmc_summary.json
# 5. the bench line itself (CPU baseline = reference Python, arbitration + fp64 ground truth, scaling proxy, S1b, S = 668, truck render, S3, voxgo)
timeout 1500 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt < /dev/null; tail -c 300 $OUT/bench_line.json; echo
UGRID_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-chunks 4 2> $OUT/bench2_err.txt < /dev/null | grep "^{" | tail -1 > $OUT/bench_2rank_shared_gpu.json
UGRID_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench8_err.txt < /dev/null | grep "^{" | tail -1 > $OUT/bench_8rank_shared_gpu.json
# 6. per-share HBM bytes of N-way deals; TA microbench; power / clock trace; fused DCVGO at 1080p; bounded DVGO at the lego size
bash tools/gpu_rank_share.sh $TAG 2>&1 | tail -3 | cut -c1-200
timeout 120 build/ab/ta_lanes > $OUT/ta_lanes.json 2> $OUT/ta_lanes.err
timeout 300 python tools/smi_trace.py $OUT/smi_trace -- python tools/smi_phases.py > $OUT/smi_trace_stdout.txt 2>&1; tail -12 $OUT/smi_trace_stdout.txt
timeout 900 python tools/bench_dcvgo.py --out $OUT/dcvgo_1080p.json 2>$OUT/dcvgo_err.txt | cut -c1-300
timeout 600 python tools/bench_dvgo.py --out $OUT/dvgo_lego_800.json 2>/dev/null | cut -c1-300
: > $OUT/train_step_s3.jsonl
for ph in 1 10001; do
  timeout 600 python tools/bench_train_step.py --steps 30 --first-step $ph 2>/dev/null | tail -1 >> $OUT/train_step_s3.jsonl
done
python - $OUT/train_step_s3.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d['tv_phase'], '%.3f ms' % d['ms_per_step'], {k: round(v, 3) for k, v in d['phases_ms'].items()}, d['survivors_M'], d.get('k0_grad_lines_touched_frac'))
PY
timeout 600 python tools/bench_voxgo_train.py --model both --steps 100 --warmup 10 > $OUT/voxgo_train.jsonl 2>/dev/null; cut -c1-260 $OUT/voxgo_train.jsonl
timeout 600 python tools/bench_voxgo_train.py --model both --steps 100 --warmup 10 --lazy-loss 1 > $OUT/voxgo_train_lazy_loss.jsonl 2>/dev/null
timeout 600 python tools/bench_voxgo_train.py --model both --steps 100 --warmup 10 --native 0 > $OUT/voxgo_train_op_by_op.jsonl 2>/dev/null
python - $OUT <<'PY'
import json, sys
for f in ("voxgo_train_lazy_loss.jsonl", "voxgo_train_op_by_op.jsonl"):
    for l in open(sys.argv[1] + "/" + f):
        d = json.loads(l); print(f, d["model"], d["workload"][-14:], "native" if d["native_step"] else "op-by-op", round(d["ms_per_step"], 4))
PY
# kernel traces of the native training steps (DVGO, DCVGO masked, S3 masked) and the host profile of the DVGO step
bash tools/gpu_r5r.sh > $OUT/voxgo_trace_print.txt 2>&1; cp gpurun_out/r5r/voxgo_train_*_kernel_stats.csv $OUT/ 2>/dev/null
bash tools/gpu_r5t.sh > $OUT/s3_trace_print.txt 2>&1; cp gpurun_out/r5t/train_step_s3_masked_kernel_stats.csv $OUT/ 2>/dev/null
bash tools/gpu_r5o.sh $TAG-host > /dev/null 2>&1; cp gpurun_out/$TAG-host/voxgo_train_host_profile_dvgo.txt $OUT/ 2>/dev/null
# 7. smoke + the whole -m gpu suite
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 3000 python -m pytest tests -m gpu -q -p no:warnings --durations=15 2>&1 | tail -30 | tee $OUT/pytest_gpu.log
ls $OUT
