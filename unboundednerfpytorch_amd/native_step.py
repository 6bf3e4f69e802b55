"""The training step of the dense-grid models and of FourierGridModel issued natively: ONE autograd node whose forward is two C calls
(include/ugrid_hip.h: ugrid_voxgo_step_sample / _forward) and whose backward is one (ugrid_voxgo_step_backward), in place of the
op-by-op step's four nodes and ~30 launches issued from Python (voxgo_model.py / fourier_model.py: TrainSampleVox / TrainSample, the
k0 GridQuery, FusedRgbnet, RenderLoss).  The C side runs the same kernels on the same sizes in the same order, so loss, outputs and every gradient are the
op-by-op step's (tests/test_gpu_voxgo_train.py, tests/test_gpu_train_step.py: bit for bit where sums have a fixed order); what changes is the host time between launches.

Only the tensors a training loop reads come back as autograd outputs (loss; mse without gradient); the per-sample arrays of the
reference's return dict are handed out detached."""
import ctypes

import torch

from . import _gradpool, _lib
from . import grid as _grid

_L = _lib.load()

if int(_L.ugrid_voxgo_step_sizeof()) != ctypes.sizeof(_lib.VoxgoStep):
    raise ImportError("native_step: _lib.VoxgoStep (%d bytes) does not mirror ugrid_voxgo_step (%d bytes) of the loaded library"
                      % (ctypes.sizeof(_lib.VoxgoStep), int(_L.ugrid_voxgo_step_sizeof())))

_WEIGHTS = ("w0", "b0", "w1", "b1", "w2", "b2")


class _CountTracker:
    """The sync-free step's only knowledge of its sample counts on the host: totals [2] of an EARLIER step, copied to page-locked memory
    behind that step's scan kernel and looked at (event query, never a wait) when a later step is issued.  They size launch grids
    (ugrid_voxgo_step.hint1 / hint2 -- the kernels themselves read the live counts on the device and loop over whatever they find) and
    reveal, one step late, a stage-2 count beyond a caller-chosen capacity."""

    def __init__(self):
        self.host, self.event, self.last, self.cap2 = None, None, (0, 0), None

    def poll(self):
        if torch.cuda.is_current_stream_capturing():      # (an event query is not allowed while a hipGraph capture is open: the last known counts)
            return self.last
        if self.event is not None and self.event.query():
            self.last = (int(self.host[0]), int(self.host[1]))
            self.event = None
            if self.cap2 is not None and self.last[1] > self.cap2:
                raise RuntimeError("sync-free training step: %d samples reached the rgbnet in an earlier step, the per-sample arrays hold %d "
                                   "(pack['sync_free']['capacity']); the samples beyond it were dropped -- raise the capacity" % (self.last[1], self.cap2))
        return self.last

    def track(self, totals, cap2):
        if self.event is not None or torch.cuda.is_current_stream_capturing():
            return
        if self.host is None:
            self.host = torch.empty(2, dtype=torch.int64).pin_memory()
        self.host.copy_(totals, non_blocking=True)
        self.cap2 = cap2
        self.event = torch.cuda.Event()
        self.event.record()


_TRACKERS = {}


def _coef9(coef):
    """ops.loss_coefficients' tuple as the 9 floats of ugrid_render_loss (an 8-tuple of an older caller: weight_freq = 0)"""
    c = [float(x) for x in coef]
    if len(c) not in (8, 9):
        raise RuntimeError("loss coefficients: 8 or 9 numbers (ops.loss_coefficients), got %d" % len(c))
    return c + [0.0] * (9 - len(c))


class VoxGOStep(torch.autograd.Function):
    """forward(density_grid [P,1,X,Y,Z], k0_grid [P,C,X,Y,Z], w0, b0, w1, b1, w2, b2, pack) -> loss, mse
    pack (dict, not differentiated): mode 'dvgo' | 'dcvgo' | 'fourier', cfg (the dict TrainSampleVox takes; 'fourier': act_shift,
    interval, thres, scene_center, scene_radius, bg_len, norm_l2, freq_num, k0_freq_num), rays_o / rays_d / viewdirs [R,3],
    viewfreq [pe], t (dcvgo, fourier: the sample table [S]), xyz_min / xyz_max, k0_xyz_min / k0_xyz_max, mask (bool [mi,mj,mk];
    none for 'fourier'), target [R,3], bg [R,3] or None, coef (ops.loss_coefficients).
    pack['sync_free'] (optional: True or {'capacity': rows of the stage-2 arrays (default rays x slots: cannot overflow), 'hints': (M1, M2)
    expected counts}): the step makes no host read and synchronises nothing -- forward and backward only enqueue work, the whole step can
    run ahead of the host or be captured in a hipGraph (include/ugrid_hip.h ugrid_voxgo_step.sync_free).  The per-sample outputs then
    have the capacity's length with pack['out']['n_valid'] (device int64 [2]) rows written.
    pack['k0_grad_ready'] (optional, may be set any time before the backward): callable(k0_parameter) invoked in the middle of the
    backward, as soon as the k0 grid's gradient is complete and assigned to .grad -- train_step.train_iteration starts the k0
    update there, beside the density half of the backward; the node then reports no gradient for the k0 grid.  On return pack['out'] holds the detached per-sample / per-ray
    arrays: alphainv_last, weights, rgb_marched, raw_alpha, raw_density, raw_logits, ray_id, step_id, t, inner, and loss_mse
    (the two scalars as one [2] tensor: a training loop that logs both reads them with one copy)."""

    @staticmethod
    def forward(ctx, density_grid, k0_grid, w0, b0, w1, b1, w2, b2, pack):
        cfg, mode = pack['cfg'], pack['mode']
        # (dense copies where a caller hands in views: the op-by-op ops do the same)
        rays_o, rays_d, viewdirs = (pack[k].contiguous() for k in ('rays_o', 'rays_d', 'viewdirs'))
        target, viewfreq = pack['target'].contiguous(), pack['viewfreq'].contiguous()
        bg = pack['bg'].contiguous() if pack.get('bg') is not None else None
        ws_ = [x.contiguous() for x in (w0, b0, w1, b1, w2, b2)]
        f32 = [("density grid", density_grid), ("rays_o", rays_o), ("rays_d", rays_d), ("viewdirs", viewdirs),
               ("target", target), ("viewfreq", viewfreq)] + [("rgbnet", x) for x in ws_]
        if bg is not None:
            f32.append(("bg", bg))
        mask = pack.get('mask')
        # the buffers handed to C as raw pointers get the checks grid.TrainSampleVox makes: the sample table and the four
        # box corners float32 / dense / on the device (dense copies are kept alive by `pack` below), the mask a dense bool [mi,mj,mk]
        box = {}
        for k in ('xyz_min', 'xyz_max', 'k0_xyz_min', 'k0_xyz_max'):
            box[k] = pack[k].contiguous()
            f32.append((k, box[k]))
        t_tab = pack.get('t')
        if t_tab is not None:
            t_tab = t_tab.contiguous()
            f32.append(("t", t_tab))
        if mask is not None and (mask.dtype != torch.bool or mask.dim() != 3):
            raise RuntimeError("VoxGOStep: mask must be a bool tensor [mi,mj,mk] (got %s, %d-D)" % (mask.dtype, mask.dim()))
        _lib.require_cuda(*f32, *([("mask", mask)] if mask is not None else []))
        _lib.require_f32(*f32, ("k0 grid", k0_grid))
        if any(x.device != density_grid.device for _, x in f32) or k0_grid.device != density_grid.device:
            raise RuntimeError("VoxGOStep: every tensor of the step must be on the density grid's device")
        _lib.wait_pending(density_grid)     # an optimizer update of a grid may still run on a side stream (step(overlap=...)); the
                                            # k0 grid's is waited for AFTER the sampling march, which does not read it
        if density_grid.dim() != 5 or density_grid.shape[1] != 1 or not density_grid.is_contiguous():
            raise RuntimeError("VoxGOStep: the density grid must be a contiguous [P,1,X,Y,Z]")
        if k0_grid.dim() != 5:
            raise RuntimeError("VoxGOStep: the k0 grid must be [P,C,X,Y,Z]")
        if mode != 'fourier' and (density_grid.shape[0] != 1 or k0_grid.shape[0] != 1):
            raise RuntimeError("VoxGOStep: the dense-grid models' grids have one level")
        k0_cl = bool(_lib.require_cuda_grid(("k0 grid", k0_grid)))
        dev = density_grid.device
        R = rays_o.shape[0]
        t = t_tab
        S = int(cfg['slots']) if mode == 'dvgo' else t.numel()
        C, W, pe = k0_grid.shape[1], ws_[0].shape[0], viewfreq.numel()
        if tuple(ws_[0].shape) != (W, C + 3 + 6 * pe) or tuple(ws_[2].shape) != (W, W) or tuple(ws_[4].shape) != (3, W):
            raise RuntimeError("VoxGOStep: rgbnet weights must be [W, C+3+6pe], [W,W], [3,W]")
        if viewdirs.shape != (R, 3) or rays_d.shape != (R, 3) or target.shape != (R, 3) or (bg is not None and bg.shape != (R, 3)):
            raise RuntimeError("VoxGOStep: rays_o, rays_d, viewdirs, target must all be [R,3]")
        key = (dev, R * S)
        sc = _grid.TrainSampleVox._scratch.get(key)
        if sc is None:
            _grid.TrainSampleVox._scratch.clear()          # one ray-batch shape at a time: 32 B per (ray, slot)
            sc = (torch.empty(R * S, 3, device=dev), torch.empty(R * S, device=dev), torch.empty(R * S, dtype=torch.int32, device=dev),
                  torch.empty(R * S, device=dev), torch.empty(R * S, device=dev))
            _grid.TrainSampleVox._scratch[key] = sc
        counts = torch.empty(2, R, dtype=torch.int32, device=dev)
        i64 = torch.empty(4 * R + 2, dtype=torch.int64, device=dev)        # offsets [2,R] | totals [2] | seg [2R]
        perray = torch.empty(R, 7, device=dev)                             # ray_tot [R,2] | partial [R,5]
        ainv = torch.empty(R, device=dev)
        rgb_marched = torch.empty(R, 3, device=dev)
        out2 = torch.empty(2, device=dev)
        s = _lib.VoxgoStep()
        s.mode = {'dvgo': 0, 'dcvgo': 1, 'fourier': 2}[mode]
        s.k0_channels_last = int(k0_cl)
        s.P, s.kP = density_grid.shape[0], k0_grid.shape[0]
        s.freq_num, s.k0_freq_num = (max(int(cfg['freq_num']), 0), max(int(cfg['k0_freq_num']), 0)) if mode == 'fourier' else (0, 0)
        s.X, s.Y, s.Z = density_grid.shape[2:]
        s.kX, s.kY, s.kZ = k0_grid.shape[2:]
        s.C, s.pe, s.width, s.slots = C, pe, W, S
        if mask is not None:
            s.mask_dims[:] = [int(x) for x in mask.shape]
            s.mask_scale[:] = cfg['mask_scale']
            s.mask_shift[:] = cfg['mask_shift']
            s.mask = mask.data_ptr()
        s.act_shift, s.interval, s.thres = float(cfg['act_shift']), float(cfg['interval']), float(cfg['thres'])
        if mode == 'dvgo':
            s.near_clip, s.far_clip, s.stepdist = float(cfg['near']), float(cfg['far']), float(cfg['stepdist'])
        else:
            s.scene_center[:] = cfg['scene_center']
            s.scene_radius[:] = cfg['scene_radius']
            s.bg_len, s.norm_l2, s.dist_thres = float(cfg['bg_len']), int(bool(cfg['norm_l2'])), float(cfg.get('dist_thres', 0.0))
            s.t_table = t.data_ptr()
        s.coef9[:] = _coef9(pack['coef'])
        s.n_rays = R
        s.density_grid, s.k0_grid = density_grid.data_ptr(), k0_grid.data_ptr()
        s.xyz_min, s.xyz_max = box['xyz_min'].data_ptr(), box['xyz_max'].data_ptr()
        s.k0_xyz_min, s.k0_xyz_max = box['k0_xyz_min'].data_ptr(), box['k0_xyz_max'].data_ptr()
        s.viewfreq = viewfreq.data_ptr()
        for n, x in zip(_WEIGHTS, ws_):
            setattr(s, n, x.data_ptr())
        s.rays_o, s.rays_d, s.viewdirs, s.target = rays_o.data_ptr(), rays_d.data_ptr(), viewdirs.data_ptr(), target.data_ptr()
        s.bg = bg.data_ptr() if bg is not None else None
        s.sc_pts, s.sc_density, s.sc_step, s.sc_w, s.sc_T = (x.data_ptr() for x in sc)
        s.counts, s.offsets = counts.data_ptr(), i64.data_ptr()
        s.totals, s.seg = i64.data_ptr() + 16 * R, i64.data_ptr() + 16 * R + 16
        s.alphainv_last, s.rgb_marched, s.out2 = ainv.data_ptr(), rgb_marched.data_ptr(), out2.data_ptr()
        s.ray_tot, s.partial = perray.data_ptr(), perray.data_ptr() + 8 * R
        ps = ctypes.addressof(s)
        sf = pack.get('sync_free')
        tracker = None
        if sf:
            # no host read: the per-sample arrays are sized by CAPACITY (stage 1: every slot of every ray; stage 2: the same unless the
            # caller bounds it), the counts stay on the device (include/ugrid_hip.h, sync_free); grids follow an earlier step's counts
            sf = sf if isinstance(sf, dict) else {}
            cap2 = int(sf.get('capacity') or R * S)
            if not (1 <= cap2 <= R * S):
                raise RuntimeError("VoxGOStep: sync_free capacity must be in [1, rays x slots = %d], got %d" % (R * S, cap2))
            tracker = _TRACKERS.setdefault((dev, mode, R, S), _CountTracker())
            h1, h2 = sf['hints'] if sf.get('hints') is not None else tracker.poll()
            s.sync_free, s.M1, s.M2 = 1, R * S, cap2
            s.hint1, s.hint2 = (min(R * S, h1 + h1 // 4 + 1024) if h1 > 0 else 0), (min(cap2, h2 + h2 // 4 + 1024) if h2 > 0 else 0)
        with _lib.guard(dev):
            st = _lib.stream_of(density_grid)
            _lib.check(_L.ugrid_voxgo_step_sample(ps, st), "voxgo_step_sample")          # the step's one host read: M1, M2 (none: sync_free)
            if tracker is not None:
                tracker.track(i64[2 * R:2 * R + 2], s.M2 if s.M2 < R * S else None)
            _lib.wait_pending(k0_grid)
            M2 = s.M2
            ws = torch.empty(int(_L.ugrid_voxgo_step_ws_floats(ps)), device=dev)
            f4 = torch.empty(4, M2, device=dev)                                # density2 | alpha2 | weights2 | t2
            ids = torch.empty(2, M2, dtype=torch.int64, device=dev)            # ray_id2 | step_id2
            logits = torch.empty(M2, 3, device=dev)
            inner = torch.ones(M2, dtype=torch.bool, device=dev) if mode == 'dcvgo' else None
            s.ws = ws.data_ptr()
            s.density2, s.alpha2, s.weights2, s.t2 = (f4.data_ptr() + 4 * M2 * i for i in range(4))
            s.ray_id2, s.step_id2 = ids.data_ptr(), ids.data_ptr() + 8 * M2
            s.inner2 = inner.data_ptr() if inner is not None else None
            s.logits = logits.data_ptr()
            _lib.check(_L.ugrid_voxgo_step_forward(ps, st), "voxgo_step_forward")
        ctx.step = s
        # everything the struct points to stays alive until the backward has been issued
        ctx.keep = (density_grid, k0_grid, ws_, rays_o, rays_d, viewdirs, target, bg, viewfreq, t, mask, box, sc, counts, i64, perray, ainv, rgb_marched, out2, ws, f4, ids, logits, inner)
        ctx.shapes = (tuple(density_grid.shape), tuple(density_grid.stride()), tuple(k0_grid.shape), tuple(k0_grid.stride()), k0_cl)
        ctx.keys = (_gradpool.key_of(density_grid), _gradpool.key_of(k0_grid))
        ctx.wshapes = [tuple(x.shape) for x in ws_]
        ctx.pack = pack
        pack['out'] = {'alphainv_last': ainv, 'weights': f4[2], 'rgb_marched': rgb_marched, 'raw_alpha': f4[1], 'raw_density': f4[0],
                       'raw_logits': logits, 'ray_id': ids[0], 'step_id': ids[1], 't': f4[3], 'inner': inner, 'loss_mse': out2}
        if s.sync_free:
            # the per-sample arrays have the capacity's length; rows [0, n_valid[1]) are written (n_valid: stage-1 / stage-2 counts, on the device)
            pack['out']['n_valid'] = i64[2 * R:2 * R + 2]
        loss, mse = out2[0], out2[1]
        ctx.mark_non_differentiable(mse)
        return loss, mse

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, g_mse):
        s = ctx.step
        (dshape, dstride, kshape, kstride, k0_cl), (dkey, kkey) = ctx.shapes, ctx.keys
        dev = ctx.keep[0].device
        g_loss = g_loss.to(torch.float32).reshape(1).contiguous()
        gw = [torch.empty(sh, device=dev) for sh in ctx.wshapes]
        g_density = _gradpool.take(dkey, dshape, dstride, dev)
        if g_density is None:
            g_density = torch.zeros(dshape, dtype=torch.float32, device=dev)
        g_k0 = _gradpool.take(kkey, kshape, kstride, dev)                  # all zero, from the last step
        if g_k0 is None:
            g_k0 = _lib.empty_like_grid(kshape, k0_cl, dev, zero=True)
        # channel-last: the scatter also marks the 256-byte lines it adds to (the optimizer's masked passes visit only those)
        touch = _gradpool.touch_for_backward(kkey, g_k0, _L) if k0_cl else None
        ps = ctypes.addressof(s)
        ws_bwd = torch.empty(int(_L.ugrid_voxgo_step_bwd_ws_floats(ps)), device=dev)
        s.grad_loss, s.ws_bwd = g_loss.data_ptr(), ws_bwd.data_ptr()
        for n, x in zip(_WEIGHTS, gw):
            setattr(s, "g_" + n, x.data_ptr())
        s.grad_density_grid, s.grad_k0_grid = g_density.data_ptr(), g_k0.data_ptr()
        s.touch = touch.data_ptr() if touch is not None else None
        ready = ctx.pack.get('k0_grad_ready')
        if ready is not None and ctx.keep[1].grad is not None:
            ready = None       # a gradient is already accumulated on the k0 grid: this one has to be ADDED by autograd, not consumed here
        with _lib.guard(dev):
            st = _lib.stream_of(g_loss)
            if ready is None:
                _lib.check(_L.ugrid_voxgo_step_backward(ps, st), "voxgo_step_backward")
                return (g_density, g_k0, *gw, None)
            # the k0 gradient first; its consumer (the optimizer's early k0 update, on a side stream) is started before the density
            # half of the backward is issued, and owns the gradient from here on
            _lib.check(_L.ugrid_voxgo_step_backward_k0(ps, st), "voxgo_step_backward_k0")
            k0_param = ctx.keep[1]
            k0_param.grad = g_k0
            del g_k0
            ready(k0_param)
            _lib.check(_L.ugrid_voxgo_step_backward_density(ps, st), "voxgo_step_backward_density")
        return (g_density, None, *gw, None)
