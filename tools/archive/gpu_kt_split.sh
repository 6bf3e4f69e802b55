cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2h -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --tune split_gather=1 > $GRAFT_REPO_ROOT/gpurun_out/r2h/log.txt 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r2h/log.txt | cut -c1-300
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r2h -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
rm -f $(find $GRAFT_REPO_ROOT/gpurun_out/r2h -name "*.db") $(find $GRAFT_REPO_ROOT/gpurun_out/r2h -name "*kernel_trace.csv")
