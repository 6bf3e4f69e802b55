#!/bin/bash
# One visit collecting the judged artefacts of round 6 on the FINAL binary (every step under its own timeout).
# usage (through gpurun): tools/gpu_final_r6.sh <tag>   -> gpurun_out/<tag>/..., then tools/collect_profiles_r6.sh <tag> here
TAG=${1:-r6final}
OUT=gpurun_out/$TAG
mkdir -p $OUT/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python -c "import bench; print(bench.device_code_sha16()); print(bench.lib_sha16())" 2>/dev/null | tail -2 > $OUT/device_code_sha16.txt
cat $OUT/device_code_sha16.txt
CODE=$(head -1 $OUT/device_code_sha16.txt)
# 1. PMC passes on the bench workload, one counter group per rocprofv3 run (groups: tools/pmc_summarize.py PMC_GROUPS + the request-size counters)
bash tools/gpu_pmc.sh $TAG "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAVES" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" > $OUT/pmc_log.txt 2>&1
tail -4 $OUT/pmc_log.txt
# 2. kernel trace + stats of the bench command, and of the truck-shaped frame
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s1 -- python $R/bench.py --steps 5 --warmup 2 --frame-pair 0 --no-cpu-baseline --no-secondary --no-proxy > $R/$OUT/prof_bench.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "s1_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/s1_kernel_stats.csv
rm -rf $OUT/prof; mkdir -p $OUT/prof
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o truck -- python $R/bench.py --scene s1b --freq 4 --stepsize 0.5 --steps 5 --warmup 2 --frame-pair 0 --no-cpu-baseline --no-secondary --no-proxy > $R/$OUT/prof_truck.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "truck_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/truck_kernel_stats.csv
rm -rf $OUT/prof; mkdir -p $OUT/prof
# (the same command with its default two frames in flight: launches of consecutive frames overlap, durations include the neighbour's work)
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s1pair -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-proxy > $R/$OUT/prof_bench_pair.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "s1pair_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/s1_two_frames_in_flight_kernel_stats.csv
rm -rf $OUT/prof
head -4 $OUT/s1_kernel_stats.csv | cut -c1-200; grep "k_shade_pc\|k_march" $OUT/truck_kernel_stats.csv | cut -c1-200
# 3. the dense TV + Adam pass alone: time, HBM bytes by request counters
timeout 200 python tools/bench_tv_adam_dense.py --tune tv_xcd=3 2>/dev/null | tail -1 | tee -a $OUT/tv_adam_dense.jsonl | cut -c1-250
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
mkdir -p profiles/r06/pmc; rm -f profiles/r06/pmc/*.csv; cp $OUT/pmc_csv/*.csv profiles/r06/pmc/
cp $OUT/tv_adam_dense_pmc.json profiles/r06/tv_adam_dense_pmc.json
cp $OUT/s1_kernel_stats.csv profiles/r06/bench_s1_kernel_stats.csv
python tools/pmc_summarize.py profiles/r06/pmc $CODE profiles/r06/pmc_summary.json $OUT/s1_kernel_stats.csv > $OUT/pmc_summary_print.txt 2>&1
cp profiles/r06/pmc_summary.json $OUT/pmc_summary.json; tail -5 $OUT/pmc_summary_print.txt
# 5. the bench line itself under the driver's command, then two / eight gloo ranks sharing the GPU (bitwise frame equality)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.txt < /dev/null; wc -c $OUT/bench_line.json; cp bench_detail.json $OUT/bench_detail.json
UGRID_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-chunks 4 2> $OUT/bench2_err.txt < /dev/null | grep "^{" | tail -1 > $OUT/bench_2rank_shared_gpu.json
UGRID_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline 2> $OUT/bench8_err.txt < /dev/null | grep "^{" | tail -1 > $OUT/bench_8rank_shared_gpu.json
python - $OUT <<'PY'
import json, sys
for n in (2, 8):
    try:
        d = json.load(open("%s/bench_%drank_shared_gpu.json" % (sys.argv[1], n)))
        print(n, "ranks sharing the GPU:", {k: v for k, v in d.items() if "equal" in k or k in ("n_gpus", "ms_per_step")}, "detail:", d.get("detail"))
    except Exception as e:
        print(n, "ranks:", e)
PY
# 6. training steps: host-counted / sync-free, per-step loss read / deferred
: > $OUT/train_steps.jsonl
for sf in 0 1; do
  timeout 600 python tools/bench_voxgo_train.py --model both --steps 100 --blocks 4 --warmup 10 --sync-free $sf --lazy-loss $sf 2>/dev/null | grep '^{' >> $OUT/train_steps.jsonl
  for ph in 1 10001; do timeout 600 python tools/bench_train_step.py --steps 30 --blocks 3 --first-step $ph --sync-free $sf --lazy-loss $sf 2>/dev/null | grep '^{' >> $OUT/train_steps.jsonl; done
done
python - $OUT/train_steps.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print(d.get("model", "s3"), d["workload"][-22:], "sync_free", d.get("sync_free"), "lazy", d.get("lazy_loss"), round(d["ms_per_step"], 4))
PY
# 7. smoke + the whole -m gpu suite in ONE process, as the driver runs it
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 3000 python -m pytest tests -x -q -m gpu -p no:warnings --durations=10 2>&1 | tail -20 | tee $OUT/pytest_gpu.log
ls $OUT
