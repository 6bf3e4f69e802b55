#!/bin/bash
# per-kernel time of the DVGO / DCVGO training steps (rocprofv3 --kernel-trace --stats), one process per (model, phase)
T=${1:-r4z}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$T
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "dvgo both" "dcvgo masked" "dcvgo dense"; do
  set -- $cfg
  rm -rf /tmp/vx_$1_$2
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vx_$1_$2 -o p -- python $GRAFT_REPO_ROOT/tools/bench_voxgo_train.py --model $1 --phase $2 --steps 40 --warmup 4 > $OUT/line_$1_$2.json 2> $OUT/err_$1_$2.txt
  f=$(find /tmp/vx_$1_$2 -name "p_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$1_$2.csv
  head -14 $OUT/kernel_stats_$1_$2.csv | cut -c1-110
done
