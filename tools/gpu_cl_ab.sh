#!/bin/bash
# A/B of the k0 storage layout in the S3 train step: per-kernel stats, raw traces deleted (they exceed gpurun's 64 MiB return limit)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cl in ${LAYOUTS:-1 0}; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cl$cl -o t -- python tools/bench_train_step.py --steps 5 --warmup 2 --channels-last $cl 2>/dev/null < /dev/null | tail -1 | cut -c1-330
  f=$(find gpurun_out/cl$cl -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -16 "$f" | cut -c1-160 > gpurun_out/cl${cl}_stats.txt; cat gpurun_out/cl${cl}_stats.txt; else echo "no stats file"; find gpurun_out/cl$cl | head; fi
  rm -rf gpurun_out/cl$cl
done
