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
