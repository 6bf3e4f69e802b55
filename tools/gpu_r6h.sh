#!/bin/bash
# round 6, visit H: which part of the sync-free step does hipStreamEndCapture choke on?
OUT=gpurun_out/r6h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_voxgo_train.py tests/test_gpu_train_scale.py -q -k "sync_free" 2>&1 | tail -12 | tee $OUT/pytest_sync_free.log
AMD_LOG_LEVEL=1 timeout 900 python tools/dbg_graph_step.py 2>&1 | tee $OUT/dbg_graph.log | tail -80
