#!/bin/bash
# round-4 visit B: consumer-priority sweep, and the new tests (long-horizon training parity vs the reference on this GPU, the
# multi-rank worker code over gloo, the reference's callers over the pybind binding, fp64 ground truth).
cd $GRAFT_REPO_ROOT
T=r4b
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
AB_STEPS=12 tools/gpu_ab.sh $T/ab build/ab/new.so build/ab/prio1.so build/ab/prio2.so build/ab/prio3.so build/ab/prio_phased3.so build/ab/new.so build/ab/dsadd_prio2.so build/ab/prio3.so build/ab/prio2.so > gpurun_out/$T/ab_stdout.txt 2>&1
BENCH_FLAGS="--tune shade_pc=1" AB_STEPS=12 tools/gpu_ab.sh $T/ab8 build/ab/new.so build/ab/prio2.so > gpurun_out/$T/ab8_stdout.txt 2>&1
BENCH_FLAGS="--tune shade_pc=0" AB_STEPS=12 tools/gpu_ab.sh $T/ab0 build/ab/new.so > gpurun_out/$T/ab0_stdout.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_train_long.py -x -q -s > gpurun_out/$T/pytest_train_long.log 2>&1
timeout 600 python -m pytest tests/test_gpu_multi.py -x -q -s > gpurun_out/$T/pytest_multi.log 2>&1
timeout 900 python -m pytest tests/test_gpu_reference_callers.py -x -q > gpurun_out/$T/pytest_refcallers.log 2>&1
timeout 900 python -m pytest tests/test_gpu_s1_scale.py -x -q -s -k "fp64 or headline" > gpurun_out/$T/pytest_fp64.log 2>&1
for f in train_long multi refcallers fp64; do echo "== $f"; tail -4 gpurun_out/$T/pytest_$f.log; done
cat gpurun_out/$T/ab/ab.txt gpurun_out/$T/ab8/ab.txt gpurun_out/$T/ab0/ab.txt
