#!/bin/bash
# round 6, visit L: (1) the hipGraph test alone / after its file's other tests, with the HIP API log's tail; (2) the re-pipelined bf16x3 kernels
OUT=gpurun_out/r6l; mkdir -p $OUT
timeout 300 python -X faulthandler -m pytest tests/test_gpu_voxgo_train.py -q -x -k "capturable" 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -15 > $OUT/graph_alone.log; tail -3 $OUT/graph_alone.log
AMD_LOG_LEVEL=3 timeout 300 python -X faulthandler -m pytest tests/test_gpu_voxgo_train.py -q -x -k "capturable and dvgo" > $OUT/graph_alone_hiplog.txt 2>&1; grep -n "Fatal" $OUT/graph_alone_hiplog.txt | head -2
python - <<'PY'
lines = open("gpurun_out/r6l/graph_alone_hiplog.txt", errors="replace").read().splitlines()
i = next((k for k, l in enumerate(lines) if "Fatal Python error" in l), len(lines))
open("gpurun_out/r6l/graph_alone_hiplog_tail.txt", "w").write("\n".join(lines[max(0, i - 120):i + 12]) + "\n")
PY
rm -f $OUT/graph_alone_hiplog.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "rgbnet" 2>&1 | tail -3 | tee $OUT/pytest_rgbnet.log
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o dvgo -- python $R/tools/bench_voxgo_train.py --model dvgo --steps 40 --sync-free 1 --lazy-loss 1 > $R/$OUT/prof_dvgo.log 2>&1 < /dev/null )
cp $(find $OUT/prof -name "dvgo_kernel_stats.csv" | head -1) $OUT/voxgo_train_dvgo_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
grep "k_lin\|k_wgrad" $OUT/voxgo_train_dvgo_kernel_stats.csv | cut -c1-60,150-260
for sf in 0 1; do timeout 600 python tools/bench_voxgo_train.py --steps 40 --sync-free $sf --lazy-loss $sf 2>&1 | grep '^{' | tee -a $OUT/voxgo_train.jsonl | cut -c1-60,240-330; done
