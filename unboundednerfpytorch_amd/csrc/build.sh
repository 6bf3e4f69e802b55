#!/bin/bash
# Build libugrid_hip.so for gfx950 (MI355X).  Usage: csrc/build.sh [extra hipcc flags]
#   UG_OUT=path        alternative output (A/B builds); UG_OBJ=dir its object directory; UG_SHADE_FLAGS / UG_MARCH_FLAGS extra -D
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include"
O=${UG_OBJ:-../../build/obj}      # UG_OBJ: separate object directory (parallel A/B builds)
mkdir -p $O
rm -f $O/ugrid_ops.o $O/ugrid_march.o $O/ugrid_shade.o $O/ugrid_train.o $O/ugrid_train_mlp.o $O/ugrid_step.o   # a failed compile must not link a stale object
OBJS="$O/ugrid_ops.o $O/ugrid_march.o $O/ugrid_shade.o $O/ugrid_train.o $O/ugrid_train_mlp.o $O/ugrid_step.o"
pids=()
hipcc $FLAGS -c ugrid_ops.hip -o $O/ugrid_ops.o "$@" &
pids+=($!)
# packed fp32 VALU (v_pk_*_f32 from the SLP vectoriser) issues at half rate on gfx950 and needs extra moves to form
# register pairs: the VALU-bound march kernel is 16 % faster without it, the shade kernel 1.3 %
hipcc $FLAGS -fno-slp-vectorize ${UG_MARCH_FLAGS} -c ugrid_march.hip -o $O/ugrid_march.o "$@" &
pids+=($!)
hipcc $FLAGS -fno-slp-vectorize ${UG_SHADE_FLAGS} -c ugrid_shade.hip -o $O/ugrid_shade.o "$@" &
pids+=($!)
hipcc $FLAGS -c ugrid_train.hip -o $O/ugrid_train.o "$@" &
pids+=($!)
hipcc $FLAGS -c ugrid_train_mlp.hip -o $O/ugrid_train_mlp.o "$@" &
pids+=($!)
hipcc $FLAGS -c ugrid_step.hip -o $O/ugrid_step.o "$@" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done   # a bare `wait` returns 0 even when a job failed
hipcc --offload-arch=gfx950 -shared -fPIC -o ${UG_OUT:-../libugrid_hip.so} $OBJS
# the fp64 twins of the drop-in ops: a library of its own (nothing on the rendering / training path loads it)
# (an A/B build -- UG_OUT set -- gets its own f64 output beside it, so that parallel A/B builds never rewrite the shipped library)
F64_OUT=${UG_OUT_F64:-${UG_OUT:+${UG_OUT%.so}_f64.so}}
hipcc $FLAGS -c ugrid_ops_f64.hip -o $O/ugrid_ops_f64.o "$@"
hipcc --offload-arch=gfx950 -shared -fPIC -o ${F64_OUT:-../libugrid_hip_f64.so} $O/ugrid_ops_f64.o
