"""unboundednerfpytorch_amd -- MI355X (gfx950) native hot path of sjtuytc/UnboundedNeRFPytorch's
FourierGrid/DVGO volumetric renderer.

Layout
  csrc/                 hand-written HIP kernels + the C ABI (include/ugrid_hip.h) -> libugrid_hip.so
  _lib.py               ctypes binding (no torch extension, no CPU fallback)
  render_utils_cuda.py, total_variation_cuda.py, ub360_utils_cuda.py, adam_upd_cuda.py
                        drop-ins for the reference's four pybind modules (same names/signatures)
  ops.py, masked_adam.py   Raw2Alpha / Raw2Alpha_nonuni / Alphas2Weights, MaskedAdam
  grid.py               grid_query, GridQuery (autograd), FourierGrid / MaskGrid module drop-ins
  fourier_render.py     FourierGridRenderer: fused march + shade render of FourierGridModel.forward
  dist.py               ray sharding + RCCL all-gather of rendered tiles
  sharded_adam.py       ShardedMaskedAdam: reduce-scatter / shard update / all-gather of the voxel grids
  compat.py             install_as_reference_extensions()

Importing a submodule loads libugrid_hip.so and raises if it is missing.
"""
__version__ = "0.1.0"

# HIP streams are served by a handful of hardware queues (4 by default); two streams that land on one queue execute in issue order.  The
# frame loops keep 3-4 views in flight on as many streams and the training step overlaps its k0 update on another: with 16 queues every set
# of streams the package creates gets queues of its own (profiles/r06/side_stream_queues.txt: S1 frame 8.12 / 8.07 ms at 3 / 4 in flight,
# no shared-queue outliers in any sweep; with 2 queues nothing overlaps).  Read by the runtime when it initialises -- i.e. effective when
# the package is imported before the first GPU call of the process, as bench.py and the tools do; an explicit setting wins.
import os as _os

_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
