#!/bin/bash
# Build libugrid_hip.so for gfx950 (MI355X).  Usage: csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I../../include \
  -o ../libugrid_hip.so ugrid_ops.hip ugrid_fused.hip "$@"
