#!/bin/bash
# per-kernel stats of the S3 train step for the k0 storage layouts in $LAYOUTS (1 = channel-last, 0 = row-major);
# raw traces are deleted (they exceed gpurun's 64 MiB return limit)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cl in ${LAYOUTS:-1 0}; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cl$cl -o t -- python tools/bench_train_step.py --steps 5 --warmup 2 --channels-last $cl 2>/dev/null < /dev/null | tail -1 | cut -c140-330
  f=$(find gpurun_out/cl$cl -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then python tools/kstats.py "$f" 60 > gpurun_out/cl${cl}_stats.txt; else echo "no stats file"; fi
  rm -rf gpurun_out/cl$cl
done
