"""Voxel-grid queries on the canonical [P,C,X,Y,Z] layout:
  grid_query     <- FourierGrid.forward (FourierGrid_grid.py:60-78) / DenseGrid.forward (grid.py:50-61), no autograd
  GridQuery      the same lookup as an autograd.Function: HIP forward + HIP scatter backward into the grid
                 (replaces F.grid_sample and its atomics-bound backward in training)
  FourierGrid    drop-in for the reference's nn.Module of that name (FourierGrid_grid.py:43-103): same constructor,
                 parameter / buffer names (state_dict compatible) and methods
  MaskGrid       <- MaskGrid.forward (grid.py:230-239, FourierGrid_grid.py:159-168)"""
import torch
import torch.nn.functional as F

from . import _gradpool, _lib, render_utils_cuda

_L = _lib.load()


@torch.no_grad()
def grid_query(grid, xyz, xyz_min, xyz_max, freq_num):
    """grid [P,C,X,Y,Z]; xyz [...,3] world coords -> [...,C] (squeezed if C==1).  freq_num=F>0: P=1+2F Fourier
    levels, mean over levels; freq_num<=0: plain dense grid (P=1).  Zero padding outside the grid."""
    cl = _lib.require_cuda_grid(("grid", grid))
    _lib.require_cuda(("xyz_min", xyz_min), ("xyz_max", xyz_max))
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
    fn = _L.ugrid_grid_query_cl if cl else _L.ugrid_grid_query
    with _lib.guard(grid.device):
        _lib.check(fn(_lib.ptr(grid), P, C, X, Y, Z, _lib.ptr(pts), _lib.ptr(xyz_min),
                                       _lib.ptr(xyz_max), max(int(freq_num), 0), pts.shape[0], _lib.ptr(out),
                                       _lib.stream_of(grid)), "grid_query")
    out = out.reshape(*lead, C)
    return out.squeeze(-1) if C == 1 else out


class GridQuery(torch.autograd.Function):
    """out[..., C] = mean over the Fourier levels of trilinear(grid[l], level coordinates of xyz).  Differentiable in
    the grid only (the reference never differentiates the sample positions: rays carry no gradient)."""

    @staticmethod
    def forward(ctx, grid, xyz, xyz_min, xyz_max, freq_num):
        _lib.wait_pending(grid)      # an optimizer update of this grid may still run on a side stream (step(overlap=...))
        out = grid_query(grid, xyz, xyz_min, xyz_max, freq_num)
        if grid.requires_grad:
            ctx.save_for_backward(xyz.reshape(-1, 3).contiguous(), xyz_min, xyz_max)
            ctx.shape = tuple(grid.shape)
            ctx.freq_num = max(int(freq_num), 0)
            ctx.channels_last = _lib.is_channels_last(grid)    # the gradient is produced in the grid's own layout
            ctx.pool_key, ctx.grid_stride = _gradpool.key_of(grid), tuple(grid.stride())
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        pts, xyz_min, xyz_max = ctx.saved_tensors
        P, C, X, Y, Z = ctx.shape
        g = grad_out.reshape(-1, C).to(torch.float32).contiguous()
        grad_grid = _gradpool.take(ctx.pool_key, ctx.shape, ctx.grid_stride, g.device)    # all zero, from the last step
        if grad_grid is None:
            grad_grid = _lib.empty_like_grid(ctx.shape, ctx.channels_last, g.device, zero=True)
        # channel-last: the scatter also marks the 256-byte lines it adds to (the optimizer's masked passes visit only those)
        touch = _gradpool.touch_for_backward(ctx.pool_key, grad_grid, _L) if ctx.channels_last else None
        with _lib.guard(g.device):
            if touch is not None:
                _lib.check(_L.ugrid_grid_query_backward_cl_touch(
                    _lib.ptr(g), P, C, X, Y, Z, _lib.ptr(pts), _lib.ptr(xyz_min), _lib.ptr(xyz_max), ctx.freq_num, pts.shape[0],
                    _lib.ptr(grad_grid), _lib.ptr(touch), _lib.stream_of(g)), "grid_query_backward (touch)")
            else:
                fn = _L.ugrid_grid_query_backward_cl if ctx.channels_last else _L.ugrid_grid_query_backward
                _lib.check(fn(_lib.ptr(g), P, C, X, Y, Z, _lib.ptr(pts), _lib.ptr(xyz_min),
                              _lib.ptr(xyz_max), ctx.freq_num, pts.shape[0],
                              _lib.ptr(grad_grid), _lib.stream_of(g)), "grid_query_backward")
        return grad_grid, None, None, None, None


class TrainMarch(torch.autograd.Function):
    """Fused stage 1 of the training forward (FourierGrid_model.py:554-598): rays -> the samples whose alpha exceeds
    fast_color_thres, compacted ray-major, with their raw densities -- sample_ray, the density lookup of all R*S points,
    Raw2Alpha, the mask and its boolean-index gathers in two kernels (ugrid_train_march / ugrid_train_compact) and ONE
    host read (the survivor count).  Differentiable in the density grid only: the backward scatters the M1 incoming
    density gradients with ugrid_grid_query_backward (the composed path runs that scatter over all R*S points).

    forward(grid [P,1,X,Y,Z], rays_o [R,3], rays_d [R,3], t [S], scene_center, scene_radius, xyz_min, xyz_max, bg_len,
            norm_l2, act_shift, interval, thres, freq_num) -> pts [M1,3], density [M1], ray_id [M1] i64, step_id [M1] i64,
            t [M1]"""
    _scratch = {}

    @staticmethod
    def forward(ctx, grid, rays_o, rays_d, t, scene_center, scene_radius, xyz_min, xyz_max, bg_len, norm_l2, act_shift,
                interval, thres, freq_num):
        import ctypes
        _lib.wait_pending(grid)      # an optimizer update of this grid may still run on a side stream (step(overlap=...))
        _lib.require_cuda(("grid", grid), ("rays_o", rays_o), ("rays_d", rays_d), ("t", t), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
        _lib.require_f32(("grid", grid), ("rays_o", rays_o), ("rays_d", rays_d), ("t", t))
        if grid.dim() != 5 or grid.shape[1] != 1:
            raise RuntimeError("density grid must be [P,1,X,Y,Z]")
        P, _, X, Y, Z = grid.shape
        R, S = rays_o.shape[0], t.numel()
        dev = grid.device
        key = (dev, R * S)
        sc = TrainMarch._scratch.get(key)
        if sc is None:
            TrainMarch._scratch.clear()          # one ray-batch shape at a time: 24 B per (ray, sample)
            sc = (torch.empty(R * S, 3, device=dev), torch.empty(R * S, device=dev), torch.empty(R * S, dtype=torch.int32, device=dev))
            TrainMarch._scratch[key] = sc
        count = torch.empty(R, dtype=torch.int32, device=dev)
        c3 = (ctypes.c_float * 3)(*[float(x) for x in scene_center])
        r3 = (ctypes.c_float * 3)(*[float(x) for x in scene_radius])
        F_ = max(int(freq_num), 0)
        with _lib.guard(dev):
            st = _lib.stream_of(grid)
            _lib.check(_L.ugrid_train_march(_lib.ptr(grid), P, X, Y, Z, F_, _lib.ptr(rays_o), _lib.ptr(rays_d), R, _lib.ptr(t), S,
                                            ctypes.cast(c3, ctypes.c_void_p), ctypes.cast(r3, ctypes.c_void_p), _lib.ptr(xyz_min),
                                            _lib.ptr(xyz_max), float(bg_len), int(bool(norm_l2)), float(act_shift), float(interval),
                                            float(thres), _lib.ptr(sc[0]), _lib.ptr(sc[1]), _lib.ptr(sc[2]), _lib.ptr(count), st),
                       "train_march")
            off = torch.cumsum(count, 0, dtype=torch.int64)
            M1 = int(off[-1].item()) if R > 0 else 0
            pts = torch.empty(M1, 3, device=dev)
            dens = torch.empty(M1, device=dev)
            ray_id = torch.empty(M1, dtype=torch.int64, device=dev)
            step_id = torch.empty(M1, dtype=torch.int64, device=dev)
            tt = torch.empty(M1, device=dev)
            if M1 > 0:
                _lib.check(_L.ugrid_train_compact(R, S, _lib.ptr(sc[0]), _lib.ptr(sc[1]), _lib.ptr(sc[2]), _lib.ptr(count),
                                                  _lib.ptr(off), _lib.ptr(t), _lib.ptr(pts), _lib.ptr(dens), _lib.ptr(ray_id),
                                                  _lib.ptr(step_id), _lib.ptr(tt), st), "train_compact")
        ctx.save_for_backward(pts, xyz_min, xyz_max)
        ctx.shape, ctx.freq_num = tuple(grid.shape), F_
        ctx.pool_key, ctx.grid_stride = _gradpool.key_of(grid), tuple(grid.stride())
        ctx.mark_non_differentiable(pts, ray_id, step_id, tt)
        return pts, dens, ray_id, step_id, tt

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_pts, g_dens, g_ray, g_step, g_t):
        pts, xyz_min, xyz_max = ctx.saved_tensors
        P, C, X, Y, Z = ctx.shape
        grad_grid = _gradpool.take(ctx.pool_key, ctx.shape, ctx.grid_stride, pts.device)
        if grad_grid is None:
            grad_grid = torch.zeros(ctx.shape, dtype=torch.float32, device=pts.device)
        if pts.shape[0] > 0:
            g = g_dens.reshape(-1, 1).to(torch.float32).contiguous()
            with _lib.guard(g.device):
                _lib.check(_L.ugrid_grid_query_backward(_lib.ptr(g), P, C, X, Y, Z, _lib.ptr(pts), _lib.ptr(xyz_min), _lib.ptr(xyz_max),
                                                        ctx.freq_num, pts.shape[0], _lib.ptr(grad_grid), _lib.stream_of(g)),
                           "grid_query_backward")
        return (grad_grid,) + (None,) * 13


class TrainSample(torch.autograd.Function):
    """Stages 1 and 2 of the training forward's sampling (FourierGrid_model.py:554-629) as ONE op: TrainMarch, Raw2Alpha,
    Alphas2Weights, the weight mask and its gathers -- include/ugrid_hip.h: ugrid_train_sample / _compact / _backward.  One
    march kernel (a ray ends where its transmittance does), one cumsum + ONE host read (M1, M2), one compaction; the backward
    is one pass over the stage-1 samples + the density lookup's scatter.  Returns the stage-2 samples
    (pts, raw density, alpha, weights, ray_id, step_id, t) and alphainv_last [R]; differentiable in the density grid through
    `weights`, `alphainv_last` and the raw `density` output."""
    _scratch = {}

    @staticmethod
    def forward(ctx, grid, rays_o, rays_d, t, scene_center, scene_radius, xyz_min, xyz_max, bg_len, norm_l2, act_shift,
                interval, thres, freq_num):
        import ctypes
        _lib.wait_pending(grid)
        _lib.require_cuda(("grid", grid), ("rays_o", rays_o), ("rays_d", rays_d), ("t", t), ("xyz_min", xyz_min), ("xyz_max", xyz_max))
        _lib.require_f32(("grid", grid), ("rays_o", rays_o), ("rays_d", rays_d), ("t", t))
        if grid.dim() != 5 or grid.shape[1] != 1:
            raise RuntimeError("density grid must be [P,1,X,Y,Z]")
        P, _, X, Y, Z = grid.shape
        R, S = rays_o.shape[0], t.numel()
        dev = grid.device
        key = (dev, R * S)
        sc = TrainSample._scratch.get(key)
        if sc is None:
            TrainSample._scratch.clear()          # one ray-batch shape at a time: 32 B per (ray, sample)
            sc = (torch.empty(R * S, 3, device=dev), torch.empty(R * S, device=dev), torch.empty(R * S, dtype=torch.int32, device=dev),
                  torch.empty(R * S, device=dev), torch.empty(R * S, device=dev))
            TrainSample._scratch[key] = sc
        counts = torch.empty(2, R, dtype=torch.int32, device=dev)
        ainv = torch.empty(R, device=dev)
        c3 = (ctypes.c_float * 3)(*[float(x) for x in scene_center])
        r3 = (ctypes.c_float * 3)(*[float(x) for x in scene_radius])
        F_ = max(int(freq_num), 0)
        with _lib.guard(dev):
            st = _lib.stream_of(grid)
            _lib.check(_L.ugrid_train_sample(_lib.ptr(grid), P, X, Y, Z, F_, _lib.ptr(rays_o), _lib.ptr(rays_d), R, _lib.ptr(t), S,
                                             ctypes.cast(c3, ctypes.c_void_p), ctypes.cast(r3, ctypes.c_void_p), _lib.ptr(xyz_min),
                                             _lib.ptr(xyz_max), float(bg_len), int(bool(norm_l2)), float(act_shift), float(interval),
                                             float(thres), *[_lib.ptr(x) for x in sc], _lib.ptr(counts[0]), _lib.ptr(counts[1]),
                                             _lib.ptr(ainv), st), "train_sample")
            off = torch.cumsum(counts, 1, dtype=torch.int64)            # [2,R] inclusive
            M1, M2 = (int(x) for x in off[:, -1].tolist()) if R > 0 else (0, 0)     # the one host read
            pts1, dens1, w1, T1 = torch.empty(M1, 3, device=dev), torch.empty(M1, device=dev), torch.empty(M1, device=dev), \
                torch.empty(M1, device=dev)
            pos2 = torch.empty(M1, dtype=torch.int32, device=dev)
            pts2, dens2, alpha2, w2, tt2 = torch.empty(M2, 3, device=dev), torch.empty(M2, device=dev), torch.empty(M2, device=dev), \
                torch.empty(M2, device=dev), torch.empty(M2, device=dev)
            ray2, step2 = torch.empty(M2, dtype=torch.int64, device=dev), torch.empty(M2, dtype=torch.int64, device=dev)
            if M1 > 0:
                _lib.check(_L.ugrid_train_sample_compact(
                    R, S, float(act_shift), float(interval), float(thres), *[_lib.ptr(x) for x in sc], _lib.ptr(counts[0]),
                    _lib.ptr(off[0]), _lib.ptr(counts[1]), _lib.ptr(off[1]), _lib.ptr(t), _lib.ptr(pts1), _lib.ptr(dens1), _lib.ptr(w1),
                    _lib.ptr(T1), _lib.ptr(pos2), _lib.ptr(pts2), _lib.ptr(dens2), _lib.ptr(alpha2), _lib.ptr(w2), _lib.ptr(ray2),
                    _lib.ptr(step2), _lib.ptr(tt2), st), "train_sample_compact")
        ctx.save_for_backward(pts1, dens1, w1, T1, pos2, counts, off, ainv, xyz_min, xyz_max)
        ctx.shape, ctx.freq_num, ctx.consts = tuple(grid.shape), F_, (float(act_shift), float(interval))
        ctx.pool_key, ctx.grid_stride = _gradpool.key_of(grid), tuple(grid.stride())
        ctx.mark_non_differentiable(pts2, alpha2, ray2, step2, tt2)
        return pts2, dens2, alpha2, w2, ainv, ray2, step2, tt2

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_pts, g_dens2, g_alpha, g_w2, g_ainv, g_ray, g_step, g_t):
        pts1, dens1, w1, T1, pos2, counts, off, ainv, xyz_min, xyz_max = ctx.saved_tensors
        P, C, X, Y, Z = ctx.shape
        dev = pts1.device
        grad_grid = _gradpool.take(ctx.pool_key, ctx.shape, ctx.grid_stride, dev)
        if grad_grid is None:
            grad_grid = torch.zeros(ctx.shape, dtype=torch.float32, device=dev)
        M1, R = pts1.shape[0], ainv.shape[0]
        if M1 > 0:
            f32 = lambda g: None if g is None else g.to(torch.float32).contiguous()
            g_dens2, g_w2, g_ainv = f32(g_dens2), f32(g_w2), f32(g_ainv)
            g1 = torch.empty(M1, 1, device=dev)
            with _lib.guard(dev):
                st = _lib.stream_of(pts1)
                _lib.check(_L.ugrid_train_sample_backward(R, ctx.consts[0], ctx.consts[1], _lib.ptr(dens1), _lib.ptr(w1), _lib.ptr(T1),
                                                          _lib.ptr(pos2), _lib.ptr(counts[0]), _lib.ptr(off[0]), _lib.ptr(ainv),
                                                          _lib.ptr(g_w2), _lib.ptr(g_ainv), _lib.ptr(g_dens2), _lib.ptr(g1), st),
                           "train_sample_backward")
                _lib.check(_L.ugrid_grid_query_backward(_lib.ptr(g1), P, C, X, Y, Z, _lib.ptr(pts1), _lib.ptr(xyz_min), _lib.ptr(xyz_max),
                                                        ctx.freq_num, M1, _lib.ptr(grad_grid), st), "grid_query_backward")
        return (grad_grid,) + (None,) * 13


class TrainSampleVox(torch.autograd.Function):
    """TrainSample for the reference's two dense-grid models: the sampling of DirectContractedVoxGO.forward (dcvgo.py:228-330,
    cfg['mode'] == 'dcvgo') and of DirectVoxGO.forward (dvgo.py:306-375, 'dvgo') as one march + one compaction
    (include/ugrid_hip.h: ugrid_train_sample_dcvgo / _dvgo / _compact_vox), differentiable in the density grid through
    `weights`, `alphainv_last` and the raw `density` output -- the backward is TrainSample's (ugrid_train_sample_backward +
    the lookup's scatter).

    forward(grid [1,1,X,Y,Z], rays_o [R,3], rays_d [R,3], t [S] or None, xyz_min, xyz_max, mask [mi,mj,mk] bool, cfg) ->
        pts [M2,3], density [M2], alpha [M2], weights [M2], alphainv_last [R], ray_id [M2] i64, step_id [M2] i64, t [M2]
        (dvgo: float(step_id)), inner [M2] bool (dvgo: all True)
    cfg (host values): mode, act_shift, interval, thres, mask_scale[3], mask_shift[3] and
        dcvgo: scene_center[3], scene_radius[3], bg_len, norm_l2, dist_thres;   dvgo: near, far, stepdist, slots"""
    _scratch = {}

    @staticmethod
    def forward(ctx, grid, rays_o, rays_d, t, xyz_min, xyz_max, mask, cfg):
        import ctypes
        _lib.wait_pending(grid)
        _lib.require_cuda(("grid", grid), ("rays_o", rays_o), ("rays_d", rays_d), ("xyz_min", xyz_min), ("xyz_max", xyz_max), ("mask", mask))
        _lib.require_f32(("grid", grid), ("rays_o", rays_o), ("rays_d", rays_d))
        if grid.dim() != 5 or grid.shape[0] != 1 or grid.shape[1] != 1 or not grid.is_contiguous():
            raise RuntimeError("density grid must be a contiguous [1,1,X,Y,Z]")
        if mask.dtype != torch.bool or mask.dim() != 3 or not mask.is_contiguous():
            raise RuntimeError("mask must be a contiguous bool [mi,mj,mk]")
        mode = cfg['mode']
        _, _, X, Y, Z = grid.shape
        R = rays_o.shape[0]
        if mode == 'dcvgo':
            _lib.require_cuda(("t", t))
            _lib.require_f32(("t", t))
            S = t.numel()
        elif mode == 'dvgo':
            S = int(cfg['slots'])
        else:
            raise ValueError(mode)
        dev = grid.device
        key = (dev, R * S)
        sc = TrainSampleVox._scratch.get(key)
        if sc is None:
            TrainSampleVox._scratch.clear()          # one ray-batch shape at a time: 32 B per (ray, slot)
            sc = (torch.empty(R * S, 3, device=dev), torch.empty(R * S, device=dev), torch.empty(R * S, dtype=torch.int32, device=dev),
                  torch.empty(R * S, device=dev), torch.empty(R * S, device=dev))
            TrainSampleVox._scratch[key] = sc
        counts = torch.empty(2, R, dtype=torch.int32, device=dev)
        ainv = torch.empty(R, device=dev)
        f3 = lambda v: (ctypes.c_float * 3)(*[float(x) for x in v])
        md = (ctypes.c_int32 * 3)(*[int(x) for x in mask.shape])
        ms, mh = f3(cfg['mask_scale']), f3(cfg['mask_shift'])
        vp = lambda a: ctypes.cast(a, ctypes.c_void_p)
        shift, interval, thres = float(cfg['act_shift']), float(cfg['interval']), float(cfg['thres'])
        with _lib.guard(dev):
            st = _lib.stream_of(grid)
            if mode == 'dcvgo':
                c3, r3 = f3(cfg['scene_center']), f3(cfg['scene_radius'])
                _lib.check(_L.ugrid_train_sample_dcvgo(
                    _lib.ptr(grid), X, Y, Z, _lib.ptr(rays_o), _lib.ptr(rays_d), R, _lib.ptr(t), S, vp(c3), vp(r3), _lib.ptr(xyz_min),
                    _lib.ptr(xyz_max), float(cfg['bg_len']), int(bool(cfg['norm_l2'])), float(cfg['dist_thres']), _lib.ptr(mask), vp(md),
                    vp(ms), vp(mh), shift, interval, thres, *[_lib.ptr(x) for x in sc], _lib.ptr(counts[0]), _lib.ptr(counts[1]),
                    _lib.ptr(ainv), st), "train_sample_dcvgo")
            else:
                _lib.check(_L.ugrid_train_sample_dvgo(
                    _lib.ptr(grid), X, Y, Z, _lib.ptr(rays_o), _lib.ptr(rays_d), R, S, _lib.ptr(xyz_min), _lib.ptr(xyz_max),
                    float(cfg['near']), float(cfg['far']), float(cfg['stepdist']), _lib.ptr(mask), vp(md), vp(ms), vp(mh), shift,
                    interval, thres, *[_lib.ptr(x) for x in sc], _lib.ptr(counts[0]), _lib.ptr(counts[1]), _lib.ptr(ainv), st),
                    "train_sample_dvgo")
            off = torch.cumsum(counts, 1, dtype=torch.int64)            # [2,R] inclusive
            M1, M2 = (int(x) for x in off[:, -1].tolist()) if R > 0 else (0, 0)     # the one host read
            pts1, dens1, w1, T1 = torch.empty(M1, 3, device=dev), torch.empty(M1, device=dev), torch.empty(M1, device=dev), \
                torch.empty(M1, device=dev)
            pos2 = torch.empty(M1, dtype=torch.int32, device=dev)
            pts2, dens2, alpha2, w2, tt2 = torch.empty(M2, 3, device=dev), torch.empty(M2, device=dev), torch.empty(M2, device=dev), \
                torch.empty(M2, device=dev), torch.empty(M2, device=dev)
            ray2, step2 = torch.empty(M2, dtype=torch.int64, device=dev), torch.empty(M2, dtype=torch.int64, device=dev)
            inner2 = torch.ones(M2, dtype=torch.bool, device=dev)
            if M1 > 0:
                _lib.check(_L.ugrid_train_sample_compact_vox(
                    R, S, shift, interval, thres, *[_lib.ptr(x) for x in sc], _lib.ptr(counts[0]), _lib.ptr(off[0]), _lib.ptr(counts[1]),
                    _lib.ptr(off[1]), _lib.ptr(t) if mode == 'dcvgo' else None, _lib.ptr(pts1), _lib.ptr(dens1), _lib.ptr(w1),
                    _lib.ptr(T1), _lib.ptr(pos2), _lib.ptr(pts2), _lib.ptr(dens2), _lib.ptr(alpha2), _lib.ptr(w2), _lib.ptr(ray2),
                    _lib.ptr(step2), _lib.ptr(tt2), _lib.ptr(inner2) if mode == 'dcvgo' else None, st), "train_sample_compact_vox")
        ctx.save_for_backward(pts1, dens1, w1, T1, pos2, counts, off, ainv, xyz_min, xyz_max)
        ctx.shape, ctx.consts = tuple(grid.shape), (shift, interval)
        ctx.pool_key, ctx.grid_stride = _gradpool.key_of(grid), tuple(grid.stride())
        ctx.mark_non_differentiable(pts2, alpha2, ray2, step2, tt2, inner2)
        return pts2, dens2, alpha2, w2, ainv, ray2, step2, tt2, inner2

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_pts, g_dens2, g_alpha, g_w2, g_ainv, g_ray, g_step, g_t, g_inner):
        pts1, dens1, w1, T1, pos2, counts, off, ainv, xyz_min, xyz_max = ctx.saved_tensors
        P, C, X, Y, Z = ctx.shape
        dev = pts1.device
        grad_grid = _gradpool.take(ctx.pool_key, ctx.shape, ctx.grid_stride, dev)
        if grad_grid is None:
            grad_grid = torch.zeros(ctx.shape, dtype=torch.float32, device=dev)
        M1, R = pts1.shape[0], ainv.shape[0]
        if M1 > 0:
            f32 = lambda g: None if g is None else g.to(torch.float32).contiguous()
            g_dens2, g_w2, g_ainv = f32(g_dens2), f32(g_w2), f32(g_ainv)
            g1 = torch.empty(M1, 1, device=dev)
            with _lib.guard(dev):
                st = _lib.stream_of(pts1)
                _lib.check(_L.ugrid_train_sample_backward(R, ctx.consts[0], ctx.consts[1], _lib.ptr(dens1), _lib.ptr(w1), _lib.ptr(T1),
                                                          _lib.ptr(pos2), _lib.ptr(counts[0]), _lib.ptr(off[0]), _lib.ptr(ainv),
                                                          _lib.ptr(g_w2), _lib.ptr(g_ainv), _lib.ptr(g_dens2), _lib.ptr(g1), st),
                           "train_sample_backward")
                _lib.check(_L.ugrid_grid_query_backward(_lib.ptr(g1), P, C, X, Y, Z, _lib.ptr(pts1), _lib.ptr(xyz_min), _lib.ptr(xyz_max),
                                                        0, M1, _lib.ptr(grad_grid), st), "grid_query_backward")
        return (grad_grid,) + (None,) * 7


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
        # config={'channels_last': True}: store a multi-channel grid as [P][X][Y][Z][C] (torch.channels_last_3d of the same
        # logical [P,C,X,Y,Z] parameter -- state_dicts / checkpoints are unchanged): the HIP lookup, its scatter backward,
        # the TV gradient and the fused TV + Adam pass then touch one 4C-byte run per voxel instead of C planes
        self.channels_last = bool(config.get('channels_last', False)) if isinstance(config, dict) else False
        if use_nerf_pos:
            self.nerf_pos_num_freq = fourier_freq_num
            self.pos_embed_output_dim = 1 + self.nerf_pos_num_freq * 2
            self.grid = torch.nn.Parameter(self._alloc([self.pos_embed_output_dim, channels, *world_size]))
        else:
            self.nerf_pos_num_freq = -1
            self.pos_embed_output_dim = -1
            self.grid = torch.nn.Parameter(self._alloc([1, channels, *world_size]))

    def _alloc(self, shape):
        shape = [int(x) for x in shape]
        if self.channels_last and shape[1] > 1:
            return torch.zeros(shape).contiguous(memory_format=torch.channels_last_3d)
        return torch.zeros(shape)

    def _as_stored(self, t):
        if self.channels_last and t.shape[1] > 1:
            return t.contiguous(memory_format=torch.channels_last_3d)
        return t.contiguous()

    # test hooks: another implementation of the lookup (differentiable, fourier_grid_query's signature) and of the
    # total_variation_cuda module; None = the HIP kernels
    query_fn = None
    tv_module = None

    def forward(self, xyz):
        """xyz [..., 3] world coordinates -> [..., C] (squeezed when C == 1)"""
        q = self.query_fn or GridQuery.apply
        _lib.wait_pending(self.grid)
        return q(self.grid, xyz, self.xyz_min, self.xyz_max, self.nerf_pos_num_freq)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        _lib.wait_pending(self.grid)
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def scale_volume_grid(self, new_world_size):
        _lib.wait_pending(self.grid)
        if self.channels == 0:
            self.grid = torch.nn.Parameter(torch.zeros([1, self.channels, *new_world_size]))
        else:
            self.grid = torch.nn.Parameter(self._as_stored(
                F.interpolate(self.grid.data, size=tuple(int(x) for x in new_world_size), mode='trilinear', align_corners=True)))

    def total_variation_add_grad(self, wx, wy, wz, dense_mode):
        """Add the total-variation gradient in place (total_variation_kernel.cu:14-67)."""
        tv = self.tv_module
        if tv is None:
            from . import total_variation_cuda as tv
        _lib.wait_pending(self.grid)
        tv.total_variation_add_grad(self.grid, self.grid.grad, wx, wy, wz, dense_mode)

    def get_dense_grid(self):
        _lib.wait_pending(self.grid)
        return self.grid

    @torch.no_grad()
    def __isub__(self, val):
        _lib.wait_pending(self.grid)
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
