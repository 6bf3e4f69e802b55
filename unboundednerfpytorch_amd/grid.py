"""Voxel-grid queries on the canonical [P,C,X,Y,Z] layout:
  grid_query     <- FourierGrid.forward (FourierGrid_grid.py:60-78) / DenseGrid.forward (grid.py:50-61), no autograd
  GridQuery      the same lookup as an autograd.Function: HIP forward + HIP scatter backward into the grid
                 (replaces F.grid_sample and its atomics-bound backward in training)
  FourierGrid    drop-in for the reference's nn.Module of that name (FourierGrid_grid.py:43-103): same constructor,
                 parameter / buffer names (state_dict compatible) and methods
  MaskGrid       <- MaskGrid.forward (grid.py:230-239, FourierGrid_grid.py:159-168)"""
import torch
import torch.nn.functional as F

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
    _lib.require_cuda(("xyz", pts))
    if pts.device != grid.device or xyz_min.device != grid.device or xyz_max.device != grid.device:
        raise RuntimeError("grid, xyz, xyz_min and xyz_max must be on the same device")
    out = torch.empty(pts.shape[0], C, dtype=torch.float32, device=grid.device)
    with torch.cuda.device(grid.device):
        _lib.check(_L.ugrid_grid_query(_lib.ptr(grid), P, C, X, Y, Z, _lib.ptr(pts), _lib.ptr(xyz_min),
                                       _lib.ptr(xyz_max), max(int(freq_num), 0), pts.shape[0], _lib.ptr(out),
                                       _lib.stream_of(grid)), "grid_query")
    out = out.reshape(*lead, C)
    return out.squeeze(-1) if C == 1 else out


class GridQuery(torch.autograd.Function):
    """out[..., C] = mean over the Fourier levels of trilinear(grid[l], level coordinates of xyz).  Differentiable in
    the grid only (the reference never differentiates the sample positions: rays carry no gradient)."""

    @staticmethod
    def forward(ctx, grid, xyz, xyz_min, xyz_max, freq_num):
        out = grid_query(grid, xyz, xyz_min, xyz_max, freq_num)
        if grid.requires_grad:
            ctx.save_for_backward(xyz.reshape(-1, 3).contiguous(), xyz_min, xyz_max)
            ctx.shape = tuple(grid.shape)
            ctx.freq_num = max(int(freq_num), 0)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        pts, xyz_min, xyz_max = ctx.saved_tensors
        P, C, X, Y, Z = ctx.shape
        g = grad_out.reshape(-1, C).to(torch.float32).contiguous()
        grad_grid = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_L.ugrid_grid_query_backward(_lib.ptr(g), P, C, X, Y, Z, _lib.ptr(pts), _lib.ptr(xyz_min),
                                                    _lib.ptr(xyz_max), ctx.freq_num, pts.shape[0],
                                                    _lib.ptr(grad_grid), _lib.stream_of(g)), "grid_query_backward")
        return grad_grid, None, None, None, None


def create_grid(type, **kwargs):
    if type == 'DenseGrid':
        return FourierGrid(**kwargs)
    raise NotImplementedError


class FourierGrid(torch.nn.Module):
    """Dense voxel grid with optional Fourier levels; mirrors FourierGrid_grid.FourierGrid (same ctor arguments,
    `grid` parameter [P,C,X,Y,Z], `xyz_min` / `xyz_max` buffers, methods), with the lookup, its backward and the
    total-variation gradient on the HIP kernels."""

    def __init__(self, channels, world_size, xyz_min, xyz_max, use_nerf_pos, fourier_freq_num, config=None):
        super().__init__()
        self.channels = channels
        self.world_size = world_size
        self.register_buffer('xyz_min', torch.Tensor(xyz_min))
        self.register_buffer('xyz_max', torch.Tensor(xyz_max))
        if use_nerf_pos:
            self.nerf_pos_num_freq = fourier_freq_num
            self.pos_embed_output_dim = 1 + self.nerf_pos_num_freq * 2
            self.grid = torch.nn.Parameter(torch.zeros([self.pos_embed_output_dim, channels, *world_size]))
        else:
            self.nerf_pos_num_freq = -1
            self.pos_embed_output_dim = -1
            self.grid = torch.nn.Parameter(torch.zeros([1, channels, *world_size]))

    # test hooks: another implementation of the lookup (differentiable, fourier_grid_query's signature) and of the
    # total_variation_cuda module; None = the HIP kernels
    query_fn = None
    tv_module = None

    def forward(self, xyz):
        """xyz [..., 3] world coordinates -> [..., C] (squeezed when C == 1)"""
        q = self.query_fn or GridQuery.apply
        return q(self.grid, xyz, self.xyz_min, self.xyz_max, self.nerf_pos_num_freq)

    def scale_volume_grid(self, new_world_size):
        if self.channels == 0:
            self.grid = torch.nn.Parameter(torch.zeros([1, self.channels, *new_world_size]))
        else:
            self.grid = torch.nn.Parameter(
                F.interpolate(self.grid.data, size=tuple(new_world_size), mode='trilinear', align_corners=True))

    def total_variation_add_grad(self, wx, wy, wz, dense_mode):
        """Add the total-variation gradient in place (total_variation_kernel.cu:14-67)."""
        tv = self.tv_module
        if tv is None:
            from . import total_variation_cuda as tv
        tv.total_variation_add_grad(self.grid, self.grid.grad, wx, wy, wz, dense_mode)

    def get_dense_grid(self):
        return self.grid

    @torch.no_grad()
    def __isub__(self, val):
        self.grid.data -= val
        return self

    def extra_repr(self):
        ws = self.world_size.tolist() if torch.is_tensor(self.world_size) else list(self.world_size)
        return f'channels={self.channels}, world_size={ws}'


class MaskGrid(torch.nn.Module):
    """Occupancy lookup (known free space); same constructor as the reference's MaskGrid (FourierGrid_grid.py:139-158,
    grid.py:205-228): either an explicit boolean `mask` with its bounds, or `path` to a checkpoint whose density grid
    is max-pooled (3x3x3), turned into alpha and thresholded at `mask_cache_thres`."""

    def __init__(self, path=None, mask_cache_thres=None, mask=None, xyz_min=None, xyz_max=None):
        super().__init__()
        if path is not None:
            st = torch.load(path, map_location='cpu', weights_only=False)
            self.mask_cache_thres = mask_cache_thres
            sd, kw = st['model_state_dict'], st['model_kwargs']
            density = F.max_pool3d(sd['density.grid'], kernel_size=3, padding=1, stride=1)
            alpha = 1 - torch.exp(-F.softplus(density + sd['act_shift']) * kw['voxel_size_ratio'])
            mask = (alpha >= self.mask_cache_thres).squeeze(0).squeeze(0)
            xyz_min, xyz_max = torch.Tensor(kw['xyz_min']), torch.Tensor(kw['xyz_max'])
        else:
            mask = mask.bool()
            xyz_min = torch.as_tensor(xyz_min, dtype=torch.float32).cpu()
            xyz_max = torch.as_tensor(xyz_max, dtype=torch.float32).cpu()
        self.register_buffer('mask', mask)
        xyz_len = xyz_max - xyz_min
        scale = (torch.Tensor(list(mask.shape)) - 1) / xyz_len
        self.register_buffer('xyz2ijk_scale', scale)
        self.register_buffer('xyz2ijk_shift', -xyz_min * scale)

    lookup_module = None    # test hook: another implementation of render_utils_cuda; None = the HIP kernels

    @torch.no_grad()
    def forward(self, xyz):
        shape = xyz.shape[:-1]
        ru = self.lookup_module or render_utils_cuda
        out = ru.maskcache_lookup(self.mask, xyz.reshape(-1, 3).contiguous(), self.xyz2ijk_scale, self.xyz2ijk_shift)
        return out.reshape(shape)

    def extra_repr(self):
        return f'mask.shape={list(self.mask.shape)}'
