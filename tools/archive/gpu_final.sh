#!/bin/bash
# One visit collecting every judged artefact of the round on the FINAL binary (every step under its own timeout).
# usage (through gpurun): tools/gpu_final.sh <tag>   -> gpurun_out/<tag>/..., then tools/collect_profiles.sh <tag> here
TAG=${1:-r2final}
OUT=gpurun_out/$TAG
mkdir -p $OUT/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import bench; print(bench.lib_sha16())" 2>/dev/null | tail -1 > $OUT/lib_sha16.txt
# 1. PMC passes, one counter group per rocprofv3 run (FETCH_SIZE and WRITE_SIZE together abort the tool)
bash tools/gpu_pmc.sh $TAG "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" > $OUT/pmc_log.txt 2>&1
# 2. kernel trace + stats of the bench command
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o s1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/$OUT/prof_bench.log 2>&1 < /dev/null )
f=$(find $OUT/prof -name "s1_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/prof/s1_kernel_stats.csv.keep
rm -rf $(find $OUT/prof -mindepth 1 -maxdepth 1 -type d) 2>/dev/null; [ -f $OUT/prof/s1_kernel_stats.csv.keep ] && mv $OUT/prof/s1_kernel_stats.csv.keep $OUT/prof/s1_kernel_stats.csv
# 3. the bench line itself (CPU baseline + S1b secondary), and the same through torch.distributed.run with one RCCL rank
timeout 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt < /dev/null
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null < /dev/null | tail -1 > $OUT/bench_dist_1rank_rccl.json
# 4. training: S3 step (default = all fused paths + overlap; then the in-order variant), per-step kernel table, TV layouts
timeout 150 python tools/bench_train_step.py 2>/dev/null < /dev/null | tail -1 > $OUT/train_step.json
timeout 150 python tools/bench_train_step.py --overlap 0 2>/dev/null < /dev/null | tail -1 > $OUT/train_step_inorder.json
EXTRA="--overlap 0" timeout 300 bash tools/gpu_step_kernels.sh > /dev/null 2>&1; cp gpurun_out/stepk/per_step.txt $OUT/train_step_kernels_per_step.txt 2>/dev/null
timeout 150 python tools/bench_tv_cl.py 2>/dev/null < /dev/null | tail -1 > $OUT/tv_adam_layouts.json
# 5. drop-in ops against the reference's own kernels (oracle/_ref travels with the snapshot)
timeout 300 python tools/bench_dropin_ops.py > $OUT/dropin_ops.txt 2>&1 < /dev/null
ls -la $OUT
