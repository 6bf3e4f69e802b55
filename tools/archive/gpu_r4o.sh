#!/bin/bash
# distribution of the long-horizon training parity statistic: the test four times, each run's JSON kept
T=${1:-r4o}
mkdir -p gpurun_out/$T
for i in 1 2 3 4; do
  timeout 400 python -m pytest tests/test_gpu_train_long.py -q -p no:warnings -m gpu --tb=short -k "320" 2>&1 | tail -3 > gpurun_out/$T/pytest_$i.log
  tail -1 gpurun_out/$T/pytest_$i.log
  cp gpurun_out/train_long_parity.json gpurun_out/$T/train_long_parity_$i.json
  python -c "
import json; d=json.load(open('gpurun_out/$T/train_long_parity_$i.json')); print({k:v for k,v in d.items() if not isinstance(v,(list,dict))})"
done
