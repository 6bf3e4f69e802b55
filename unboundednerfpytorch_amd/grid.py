"""Inference-time voxel-grid queries on the canonical [P,C,X,Y,Z] layout:
  grid_query     <- FourierGrid.forward (FourierGrid_grid.py:60-78) / DenseGrid.forward (grid.py:50-61)
  MaskGrid       <- MaskGrid.forward (grid.py:230-239, FourierGrid_grid.py:159-168)
No autograd: training keeps using the reference's F.grid_sample modules."""
import torch

from . import _lib, render_utils_cuda

_L = _lib.load()


@torch.no_grad()
def grid_query(grid, xyz, xyz_min, xyz_max, freq_num):
    """grid [P,C,X,Y,Z]; xyz [...,3] world coords -> [...,C] (squeezed if C==1).  freq_num=F>0: P=1+2F Fourier
    levels, mean over levels; freq_num<=0: plain dense grid (P=1).  Zero padding outside the grid."""
    _lib.require_cuda(("grid", grid), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
    _lib.require_f32(("grid", grid), ("xyz", xyz), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
    if grid.dim() != 5:
        raise RuntimeError("grid must be [P,C,X,Y,Z]")
    P, C, X, Y, Z = grid.shape
    lead = xyz.shape[:-1]
    pts = xyz.reshape(-1, 3).contiguous()
    out = torch.empty(pts.shape[0], C, dtype=torch.float32, device=grid.device)
    with torch.cuda.device(grid.device):
        _lib.check(_L.ugrid_grid_query(_lib.ptr(grid), P, C, X, Y, Z, _lib.ptr(pts), _lib.ptr(xyz_min),
                                       _lib.ptr(xyz_max), max(int(freq_num), 0), pts.shape[0], _lib.ptr(out),
                                       _lib.stream_of(grid)), "grid_query")
    out = out.reshape(*lead, C)
    return out.squeeze(-1) if C == 1 else out


class MaskGrid(torch.nn.Module):
    """Occupancy lookup (known free space).  Built from an explicit mask (the `path=` constructor of the
    reference belongs to the checkpoint row)."""

    def __init__(self, mask, xyz_min, xyz_max):
        super().__init__()
        mask = mask.bool()
        xyz_min = torch.as_tensor(xyz_min, dtype=torch.float32)
        xyz_max = torch.as_tensor(xyz_max, dtype=torch.float32)
        self.register_buffer('mask', mask)
        xyz_len = xyz_max - xyz_min
        self.register_buffer('xyz2ijk_scale', (torch.Tensor(list(mask.shape)) - 1) / xyz_len)
        self.register_buffer('xyz2ijk_shift', -xyz_min * self.xyz2ijk_scale)

    @torch.no_grad()
    def forward(self, xyz):
        shape = xyz.shape[:-1]
        out = render_utils_cuda.maskcache_lookup(self.mask, xyz.reshape(-1, 3).contiguous(), self.xyz2ijk_scale,
                                                 self.xyz2ijk_shift)
        return out.reshape(shape)
