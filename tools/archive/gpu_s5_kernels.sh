cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s5prof -o s5 -- python $GRAFT_REPO_ROOT/tools/bench_s5_blocks.py --steps 5 --warmup 2 > /dev/null 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/s5prof -name "s5_kernel_stats.csv" | head -1)
[ -n "$f" ] && python tools/kstats.py "$f" 8
rm -rf gpurun_out/s5prof
