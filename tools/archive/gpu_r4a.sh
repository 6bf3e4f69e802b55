#!/bin/bash
# round-4 visit A: A/B of the first shade / march arms, stall + L1 counters on the default build, FETCH_SIZE calibration,
# fp64 ground truth of the S1 tail, the tests touched so far.
cd $GRAFT_REPO_ROOT
T=r4a
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/$T/device.txt 2>&1
AB_STEPS=12 tools/gpu_ab.sh $T/ab build/ab/r03.so build/ab/new.so build/ab/dsadd.so build/ab/prio2.so build/ab/dsadd_prio2.so build/ab/fcell.so build/ab/nbl4.so build/ab/new.so build/ab/r03.so > gpurun_out/$T/ab_stdout.txt 2>&1
# counters on the default (in-tree) build
tools/gpu_pmc.sh $T/pmc \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
  "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAVES" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" > gpurun_out/$T/pmc_stdout.txt 2>&1
# FETCH_SIZE calibration
mkdir -p gpurun_out/$T/fetch
i=0
for ctrs in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/fc_$i -o p -- $GRAFT_REPO_ROOT/build/ab/fetch_calib > $GRAFT_REPO_ROOT/gpurun_out/$T/fetch/fetch_calib_$i.out 2>&1 )
  f=$(find /tmp/fc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/$T/fetch/fetch_pass_$i.csv
  f=$(find /tmp/fc_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/$T/fetch/kernel_trace_$i.csv
done
grep requested_bytes gpurun_out/$T/fetch/fetch_calib_1.out > gpurun_out/$T/fetch/fetch_calib.json
python tools/microbench/fetch_calib_report.py gpurun_out/$T/fetch gpurun_out/$T/microbench_fetch_calib.json > gpurun_out/$T/fetch/report.txt 2>&1
# tests
timeout 900 python -m pytest tests/test_gpu_s1_scale.py -x -q -s -k "fp64 or headline" > gpurun_out/$T/pytest_fp64.log 2>&1
timeout 600 python -m pytest tests/test_gpu_touch.py tests/test_gpu_fused.py -x -q > gpurun_out/$T/pytest_touch_fused.log 2>&1
tail -3 gpurun_out/$T/pytest_fp64.log gpurun_out/$T/pytest_touch_fused.log
cat gpurun_out/$T/ab/ab.txt
