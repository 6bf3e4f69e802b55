#!/bin/bash
# per-step kernel table of the S3 train step (difference of two runs), raw traces deleted
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stepk
for n in 4 18; do
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stepk/n$n -o t -- python tools/bench_train_step.py --steps $n --warmup 2 $EXTRA 2>/dev/null < /dev/null | tail -1 | cut -c140-330
  f=$(find gpurun_out/stepk/n$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/stepk/n$n.csv
  rm -rf gpurun_out/stepk/n$n
done
cd tools && python kstats_diff.py ../gpurun_out/stepk/n4.csv 4 ../gpurun_out/stepk/n18.csv 18 > ../gpurun_out/stepk/per_step.txt
