#!/bin/bash
# round 6, visit K: the GPU suite file by file with whole logs kept (visit I's run died in a core dump whose head the tail cut off)
OUT=gpurun_out/r6k; mkdir -p $OUT
for f in tests/test_gpu_*.py; do
  b=$(basename $f .py)
  timeout 900 python -X faulthandler -m pytest $f -m gpu -q -x 2>&1 | grep -v "Warning\|warnings.warn\|^$" > $OUT/$b.log
  echo "$b: $(grep -E "passed|failed|error|Fatal|dumped" $OUT/$b.log | tail -2 | tr '\n' ' ')"
done | tee $OUT/summary.txt
