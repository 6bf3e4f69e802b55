#!/bin/bash
# Parity study on the S1 frame: which arithmetic deviation of k_march moves which rays.  Build container:
#   tools/gpu_parity_ab.sh build        -> build/ab/lib_<variant>.so
# GPU box:
#   tools/gpu_parity_ab.sh run          -> gpurun_out/parity_ab.txt
VARIANTS="base:  exactdiv:-DUG_EXACT_DIV  libmsincos:-DUG_LIBM_SINCOS  libmalpha:-DUG_LIBM_ALPHA  cornersum:-DUG_CORNER_SUM  all:-DUG_EXACT_DIV,-DUG_LIBM_SINCOS,-DUG_LIBM_ALPHA,-DUG_CORNER_SUM"
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build/ab
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
    UG_OUT=../../build/ab/lib_$name.so UG_MARCH_FLAGS="$flags" bash unboundednerfpytorch_amd/csrc/build.sh 2>&1 | grep -E "error" 
    ls -la build/ab/lib_$name.so | awk '{print $5, $9}'
  done
else
  mkdir -p gpurun_out
  for v in $VARIANTS; do
    name=${v%%:*}
    UGRID_LIB=build/ab/lib_$name.so python tools/gpu_parity_ab.py $name 2>&1 | grep -v -i warn
  done | tee gpurun_out/parity_ab.txt
fi
