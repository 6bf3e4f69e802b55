"""FourierGridRenderer: the inference forward of the reference's FourierGridModel
(/root/reference/FourierGrid/FourierGrid_model.py:554-672) as two fused HIP kernels.

Host-side mirror of the reference interface: `forward(rays_o, rays_d, viewdirs, **render_kwargs)`
takes the same arguments (`stepsize`, `render_depth`, ...) and returns a dict with the keys the
render program consumes -- `rgb_marched [R,3]`, `depth [R]`, `alphainv_last [R]`
(run_render.py:46).  The training-only per-sample keys (`weights`, `raw_rgb`, `ray_id`, ...) are not
materialised by the fused path; training goes through the drop-in ops (ops.py) instead.

Device data owned by the renderer:
  density bricks  [P*(G-1)^3][8]                  32 B per trilinear cell: its 8 trilinear-polynomial coefficients
  k0 bricks       [P*(G-1)^3][2][C/4][8][2]       384 B per cell at C = 12, same coefficient form per channel
  packed rgbnet   fp32 | bf16x3 | fp16x2 images   (MFMA A-operand order, 321 KB; one of them is staged in LDS)
  t / s tables    [S]
  work list       worst-case survivor list per 64-ray tile (ugrid_render_ws_bytes)
"""
import ctypes
import math

import os

import torch

from . import _lib

_L = _lib.load()
_p = _lib.ptr


def tune(key, value):
    """Tuning knobs (ugrid_tune, include/ugrid_hip.h): speed only -- 'march_waves' 4..6, 'tv_xcd' 0|1|2|3, 'shade_pc' 0|1|2; and one choice of
    arithmetic -- 'train_mlp' 0|1 (training rgbnet products on fp32 MFMAs | bf16x3, both fp32-accurate)."""
    _lib.check(_L.ugrid_tune(key.encode(), int(value)), "ugrid_tune(%s)" % key)


# UGRID_TUNE="key=value,key=value": speed knobs applied when the module is imported (A/B runs of tests / tools without code changes)
# A malformed or rejected entry is reported and skipped: an environment variable must not make the package unimportable.
def _apply_env_tune(spec):
    import warnings
    applied = []
    for kv in filter(None, (x.strip() for x in spec.split(","))):
        key, sep, val = kv.partition("=")
        try:
            if not sep:
                raise ValueError("expected key=value")
            tune(key.strip(), int(val))
            applied.append(key.strip())
        except (ValueError, RuntimeError) as e:
            warnings.warn("UGRID_TUNE entry %r ignored: %s" % (kv, e))
    return applied


_apply_env_tune(os.environ.get("UGRID_TUNE", ""))


def sample_table(world_len, stepsize, bg_len, t_boundary=1.5):  # noqa: E302  (dcvgo.py:243-250 uses t_boundary = 2)
    """Sample distances t [S] and s = 1 - 1/(1+t), computed with the same torch ops as the reference
    (FourierGrid_model.py:524-532,649) on the host; shared by every ray."""
    n_inner = int(2 / (2 + 2 * bg_len) * world_len / stepsize) + 1
    b_in = torch.linspace(0, t_boundary, n_inner + 1)
    b_out = t_boundary / torch.linspace(1, 1 / 128, n_inner + 1)
    t = torch.cat([(b_in[1:] + b_in[:-1]) * 0.5, (b_out[1:] + b_out[:-1]) * 0.5])
    s = 1 - 1 / (1 + t)
    return t, s


def rgbnet_fits_fused(ws):
    """rgbnet_depth = 3 with any width up to 128 (configs/default.py:104-105; free_dataset/*.py: width 64)"""
    return (len(ws) == 3 and ws[0].dim() == 2 and ws[1].dim() == 2 and ws[1].shape[0] == ws[1].shape[1] == ws[0].shape[0] <= 128
            and tuple(ws[2].shape) == (3, ws[1].shape[0]))


def pad_rgbnet_to_128(ws, bs):
    """A depth-3 rgbnet of width w < 128 as the 128-wide network the shade kernels are built for: units w..127 get zero weights
    and zero biases, so they output relu(0) = 0 and contribute exact zeros to the next layer's sums -- the function is unchanged."""
    w = ws[1].shape[0]
    if w == 128:
        return list(ws), list(bs)
    w0 = ws[0].new_zeros(128, ws[0].shape[1]); w0[:w] = ws[0]
    w1 = ws[1].new_zeros(128, 128); w1[:w, :w] = ws[1]
    w2 = ws[2].new_zeros(3, 128); w2[:, :w] = ws[2]
    b0 = bs[0].new_zeros(128); b0[:w] = bs[0]
    b1 = bs[1].new_zeros(128); b1[:w] = bs[1]
    return [w0, w1, w2], [b0, b1, bs[2]]


class FourierGridRenderer:
    """Fused render of a trained FourierGridModel.

    state: dict with
      density_grid [P,1,G,G,G], k0_grid [P,C,G,G,G] (or [1,3,G,G,G] when there is no rgbnet),
      rgbnet_weights / rgbnet_biases: lists of 3 nn.Linear tensors (or empty),
      scene_center[3], scene_radius[3], xyz_min[3], xyz_max[3] (contracted bounds, fp32 tensors),
      bg_len, fourier_freq_num, viewbase_pe, act_shift, voxel_size_ratio, fast_color_thres,
      contracted_norm ('inf' | 'l2'), world_len.
    Optional `dcvgo` = {'mask' bool [mx,my,mz], 'xyz2ijk_scale' [3], 'xyz2ijk_shift' [3]} with fourier_freq_num = 0
    (single-level grids): the DirectContractedVoxGO forward (dcvgo.py:228-384) -- sample table with boundary 2, the
    cumdist_thres rule and the mask cache inside the march (ugrid_render_march_dcvgo), `wsum_mid` among the outputs and
    `bg` honoured (rgb_marched += alphainv_last * bg).  dcvgo_render.DirectContractedVoxGORenderer.render_rays builds it.
    Optional `dvgo` = {'mask', 'xyz2ijk_scale', 'xyz2ijk_shift', 'voxel_size'} with fourier_freq_num = 0: the BOUNDED
    DirectVoxGO forward (dvgo.py:306-425, rgbnet_direct) -- xyz_min / xyz_max are the scene box, rays are clipped against it
    and marched with their own step counts inside ugrid_render_march_dvgo (no contraction, no sample table); render kwargs
    `near` and `bg` as the reference, depth = sum w * step_id.  dvgo_render.DirectVoxGORenderer.render_rays builds it.
    """

    def __init__(self, state, device, max_ws_bytes=48 << 30, pipeline=0, mlp_mode=None):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("FourierGridRenderer needs a HIP device (no CPU path)")
        self.device = dev
        self.max_ws_bytes = int(max_ws_bytes)
        # march kernel -> work list -> shade kernel.  (A single persistent launch that marched and shaded per wave was measured
        # slower in rounds 1-3 and left the library in round 5: tools/experiments/ARMS.md.)
        # pipeline=N>1: N ray chunks software-pipelined over two streams (measured slower: both kernels are bound by the same
        # per-CU vector-memory path), experimental.
        self.mlp_mode = _lib.MLP_BF16X3   # rgbnet arithmetic; set from ugrid_pack_mlp's answer below
        self.pipeline = int(pipeline)
        dg = state["density_grid"].to(dev, torch.float32).contiguous()
        kg = state["k0_grid"].to(dev, torch.float32).contiguous()
        self.F = int(state["fourier_freq_num"])
        P = 1 + 2 * self.F
        if dg.shape[0] != P or dg.shape[1] != 1:
            raise RuntimeError("density_grid must be [1+2F, 1, X, Y, Z]")
        self.G = tuple(int(x) for x in dg.shape[2:])
        if tuple(kg.shape[2:]) != self.G:
            raise RuntimeError("density and k0 grids must share a resolution in the fused path")
        self.has_mlp = len(state["rgbnet_weights"]) > 0
        self.C = int(kg.shape[1])
        self.pe = int(state["viewbase_pe"])
        self.bg_len = float(state["bg_len"])
        self.world_len = int(state["world_len"])
        self.thres = float(state["fast_color_thres"])
        self.act_shift = float(state["act_shift"])
        self.voxel_size_ratio = float(state["voxel_size_ratio"])
        self.norm_l2 = {"inf": 0, "l2": 1}[state.get("contracted_norm", "inf")]
        self._vec = {k: [float(v) for v in state[k]] for k in ("scene_center", "scene_radius", "xyz_min", "xyz_max")}
        self.dc = None
        if state.get("dcvgo") is not None:
            if self.F != 0:
                raise RuntimeError("the DirectContractedVoxGO march is for single-level grids (fourier_freq_num = 0)")
            d = state["dcvgo"]
            self.dc = {"mask": d["mask"].to(dev).to(torch.bool).contiguous(),
                       "scale": [float(x) for x in d["xyz2ijk_scale"]], "shift": [float(x) for x in d["xyz2ijk_shift"]]}
            if self.dc["mask"].dim() != 3:
                raise RuntimeError("dcvgo mask must be a [mx,my,mz] bool grid")
        self.dv = None
        if state.get("dvgo") is not None:
            if self.F != 0 or self.dc is not None:
                raise RuntimeError("the DirectVoxGO march is for single-level grids (fourier_freq_num = 0), without 'dcvgo'")
            d = state["dvgo"]
            self.dv = {"mask": d["mask"].to(dev).to(torch.bool).contiguous(), "voxel_size": d["voxel_size"],
                       "scale": [float(x) for x in d["xyz2ijk_scale"]], "shift": [float(x) for x in d["xyz2ijk_shift"]]}
            if self.dv["mask"].dim() != 3:
                raise RuntimeError("dvgo mask must be a [mx,my,mz] bool grid")
        elif self.F == 0 and self.dc is None:
            raise RuntimeError("fourier_freq_num = 0 is the DirectContractedVoxGO / DirectVoxGO path: pass state['dcvgo'] "
                               "or state['dvgo']")
        if self.thres <= 0:
            raise RuntimeError("fast_color_thres must be > 0 (the reference forward is not usable at 0 either, "
                               "FourierGrid_model.py:600-614)")
        with _lib.guard(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            X, Y, Z = self.G
            self.density_bricks = torch.empty(_L.ugrid_brick_bytes(P, 1, X, Y, Z, 0) // 4, dtype=torch.float32, device=dev)
            _lib.check(_L.ugrid_pack_bricks(_p(dg), P, 1, X, Y, Z, 0, _p(self.density_bricks), st), "pack density")
            if self.has_mlp:
                if kg.shape[0] != P:
                    raise RuntimeError("k0_grid must have 1+2F levels when an rgbnet is present")
                self.k0_bricks = torch.empty(_L.ugrid_brick_bytes(P, self.C, X, Y, Z, 0) // 4, dtype=torch.float32, device=dev)
                _lib.check(_L.ugrid_pack_bricks(_p(kg), P, self.C, X, Y, Z, 0, _p(self.k0_bricks), st), "pack k0")
                ws_, bs_ = state["rgbnet_weights"], state["rgbnet_biases"]
                if not rgbnet_fits_fused(ws_):
                    raise RuntimeError("fused shade supports rgbnet_depth=3, rgbnet_width <= 128 "
                                       "(configs/default.py:104-105)")
                ws_, bs_ = pad_rgbnet_to_128([x.to(dev, torch.float32) for x in ws_], [x.to(dev, torch.float32) for x in bs_])
                # residual colour (DirectVoxGO with rgbnet_direct = False, dvgo.py:385-398): the network reads [k0[3:], embedding] and
                # k0[:3] is added to its output.  The kernels gather all C channels anyway: the first layer gets ZERO columns for
                # the three diffuse channels (so the matrix chain ignores them) and the shade epilogue adds them to the logits
                self.residual = bool(state.get("rgbnet_residual", False))
                if self.residual:
                    if self.C < 9:
                        raise RuntimeError("residual colour needs k0 channels >= 9 (3 diffuse + the rgbnet's features)")
                    ws_[0] = torch.cat([ws_[0].new_zeros(ws_[0].shape[0], 3), ws_[0]], dim=1)
                self.mlp_in = int(ws_[0].shape[1])
                if self.mlp_in != self.C + 3 + 6 * self.pe:
                    raise RuntimeError("rgbnet input width must be C + 3 + 6*viewbase_pe")
                t = [x.to(dev, torch.float32).contiguous() for x in (ws_[0], bs_[0], ws_[1], bs_[1], ws_[2], bs_[2])]
                self.mlp_packed = torch.empty(_L.ugrid_mlp_packed_bytes(self.C, self.pe) // 4, dtype=torch.float32, device=dev)
                # |k0 feature| <= max |k0 grid value| (convex combinations, then a mean over levels): sizes the
                # activation scales of the fp16x2 rgbnet image; the library reports whether that mode is usable
                best = ctypes.c_int32(_lib.MLP_BF16X3)
                _lib.check(_L.ugrid_pack_mlp(*[_p(x) for x in t], self.C, self.pe, 128, float(kg.abs().max()),
                                             _p(self.mlp_packed), ctypes.byref(best), st), "pack mlp")
                self.mlp_mode = int(best.value) if mlp_mode is None else int(mlp_mode)
                if self.mlp_mode == _lib.MLP_FP16X2 and best.value != _lib.MLP_FP16X2:
                    raise RuntimeError("fp16x2 rgbnet arithmetic is not usable for this network (operand range)")
            else:
                if kg.shape[0] != 1 or self.C != 3:
                    raise RuntimeError("without an rgbnet k0 must be a single-level 3-channel grid")
                self.mlp_in = 0
                self.k0_bricks = torch.empty(_L.ugrid_brick_bytes(1, 3, X, Y, Z, 1) // 4, dtype=torch.float32, device=dev)
                _lib.check(_L.ugrid_pack_bricks(_p(kg), 1, 3, X, Y, Z, 1, _p(self.k0_bricks), st), "pack k0")
                self.mlp_packed = None
            torch.cuda.current_stream(dev).synchronize()  # dg/kg temporaries may now be freed
        self._tables = {}
        self._ws = None
        self._ws_ring = None
        self._streams = None
        self._last = None

    # -- helpers ---------------------------------------------------------------------------------
    def stepdist(self, stepsize):
        # python float * 0-d fp32 tensor -> fp32 product (dvgo.py:319)
        return float(stepsize * torch.as_tensor(self.dv["voxel_size"], dtype=torch.float32))

    def tables(self, stepsize):
        key = float(stepsize)
        if self.dv is not None:
            # no sample table: S = an upper bound of a ray's step count, ceil(box diagonal / stepdist) + 1 (sizes the work list)
            ext = [self._vec["xyz_max"][i] - self._vec["xyz_min"][i] for i in range(3)]
            diag = (ext[0] ** 2 + ext[1] ** 2 + ext[2] ** 2) ** 0.5
            return None, None, int(diag * (1 + 1e-5) / self.stepdist(stepsize)) + 2
        if key not in self._tables:
            t, s = sample_table(self.world_len, key, self.bg_len, t_boundary=2 if self.dc is not None else 1.5)
            self._tables[key] = (t.to(self.device), s.to(self.device), int(t.numel()))
        return self._tables[key]

    def interval(self, stepsize):
        # python float * 0-d fp32 tensor -> fp32 product (FourierGrid_model.py:572)
        return float(torch.tensor(self.voxel_size_ratio, dtype=torch.float32) * stepsize)

    def _params(self, n_rays, S, stepsize):
        p = _lib.RenderParams()
        p.n_rays, p.n_samples, p.freq_num = n_rays, S, self.F
        p.grid_x, p.grid_y, p.grid_z = self.G
        p.k0_channels, p.mlp_in, p.mlp_width = self.C, self.mlp_in, 128
        p.viewbase_pe, p.norm_l2 = self.pe, self.norm_l2
        for k in ("scene_center", "scene_radius", "xyz_min", "xyz_max"):
            for i in range(3):
                getattr(p, k)[i] = self._vec[k][i]
        p.bg_len = self.bg_len
        p.act_shift, p.interval, p.thres = self.act_shift, self.interval(stepsize), self.thres
        p.mlp_mode = self.mlp_mode | (_lib.MLP_RESIDUAL if getattr(self, "residual", False) else 0)
        return p

    def rays_per_chunk(self, S):
        per_ray = 17 * S + 8   # work-list entry {p, w} 16 B + ray slot 1 B per sample, worst case = ugrid_render_ws_bytes per ray
                               # (+ 256-byte alignment of the three regions and the counters: covered by the 8)
        n = max(64, (self.max_ws_bytes // per_ray) // 64 * 64)
        return n

    frames_in_flight = 3      # run_render.render_viewpoints' default for this renderer (three views on three streams / work lists: 2 % faster than two at
                              # 1080p, four is no better; profiles/r06/frames_in_flight_sweep.txt)

    def use_workspace_slot(self, k):
        """Frames in flight on different streams need work lists of their own -- the march of one fills its list while the shade of the
        other drains its (bench.py --frame-pair; run_render's views are independent frames): slot k's workspace becomes the current one."""
        slots = self.__dict__.setdefault("_ws_slots", {})
        cur = self.__dict__.get("_ws_slot", 0)
        if k != cur:
            slots[cur] = self._ws
            self._ws = slots.get(k)
            self._ws_slot = k

    def _workspace(self, n_rays, S):
        need = _L.ugrid_render_ws_bytes(n_rays, S)
        if self._ws is None or self._ws.numel() < need:
            if self._ws is not None and self._ws.is_cuda:      # (a work list may have been used on another stream than the one it was allocated on)
                self._ws.record_stream(torch.cuda.current_stream(self.device))
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # -- ray order ---------------------------------------------------------------------------------
    # The march kernel gives a wave 64 CONSECUTIVE rays of the list; when those are neighbouring pixels they hit 1-6 grid
    # cells per load and share cache lines, and the kernel runs at L1 / VALU speed.  The reference API accepts any ray list;
    # 64 unrelated rays per wave make every lane fetch its own bricks from HBM: 8.6x slower on the S1 frame
    # (profiles/r02/bench_s1_shuffled_rays.json).  forward(..., ray_order=...) therefore takes
    #   "coherent"  the caller vouches for the order (render_view, bench.py: 8 x 8 pixel blocks) -- no check, no sync;
    #   "sort"      sort the rays into direction / origin Morton order, render, put the results back (all on the device);
    #   "auto"      (default) measure the coherence of a sample of 64-ray tiles (one small reduction + one host read),
    #               then behave like "coherent" or "sort" (+ a one-time warning).  Lists under 4096 rays are not checked.
    INCOHERENT_TILE_SPREAD = 0.08     # max |viewdir - viewdir of the tile's first ray|_inf; an 8 x 8 pixel block at 1080p: ~0.005
    _warned_incoherent = False

    @staticmethod
    def tile_spread(rays_o, viewdirs, n_tiles=256):
        """mean over a sample of 64-ray tiles of the largest direction + origin deviation from the tile's first ray (0-d tensor)"""
        R = rays_o.shape[0]
        T = R // 64
        pick = torch.linspace(0, T - 1, min(n_tiles, T), device=rays_o.device).long()
        idx = (pick[:, None] * 64 + torch.arange(64, device=rays_o.device)[None, :]).reshape(-1)
        v = viewdirs.index_select(0, idx).view(-1, 64, 3)
        o = rays_o.index_select(0, idx).view(-1, 64, 3)
        dev_v = (v - v[:, :1]).abs().amax(dim=(1, 2))
        dev_o = (o - o[:, :1]).abs().amax(dim=(1, 2))
        return (dev_v + dev_o).mean()

    def morton_ray_order(self, rays_o, viewdirs):
        """permutation that puts rays with similar origin (coarse) and direction (fine) next to each other: int64 keys =
        origin cell (4 bits per axis of the normalised origin) | 2 x 12-bit Morton code of the octahedral direction map"""
        c = torch.tensor(self._vec["scene_center"], device=rays_o.device)
        r = torch.tensor(self._vec["scene_radius"], device=rays_o.device)
        on = ((rays_o - c) / r).clamp(-1.0, 1.0)
        oc = ((on + 1.0) * 7.999).long()                                        # 0..15 per axis
        v = viewdirs / viewdirs.abs().sum(-1, keepdim=True).clamp_min(1e-20)    # octahedral projection
        uv = v[:, :2]
        fold = (1.0 - uv.abs().flip(-1)) * torch.where(uv >= 0, 1.0, -1.0)
        uv = torch.where(v[:, 2:3] < 0, fold, uv)
        q = ((uv + 1.0) * 2047.999).long().clamp_(0, 4095)                      # 12 bits each

        def spread(x):      # 12 bits -> every other bit of 24
            x = (x | (x << 8)) & 0x00FF00FF
            x = (x | (x << 4)) & 0x0F0F0F0F
            x = (x | (x << 2)) & 0x33333333
            return (x | (x << 1)) & 0x55555555
        key = (((oc[:, 0] << 8) | (oc[:, 1] << 4) | oc[:, 2]) << 24) | (spread(q[:, 0]) << 1) | spread(q[:, 1])
        return torch.sort(key).indices

    # -- the reference-shaped entry point ----------------------------------------------------------
    @torch.no_grad()
    def forward(self, rays_o, rays_d, viewdirs, global_step=None, is_train=False, **render_kwargs):
        """Volume rendering of R rays.  render_kwargs: stepsize (required), render_depth (depth is always
        produced by the fused kernel; the key is returned when requested, like the reference), ray_order
        ("auto" | "coherent" | "sort", see above)."""
        if is_train or global_step is not None:
            raise RuntimeError("the fused renderer is inference-only; train through unboundednerfpytorch_amd.ops")
        order = render_kwargs.get("ray_order", "auto")
        if order not in ("auto", "coherent", "sort"):
            raise ValueError("ray_order must be 'auto', 'coherent' or 'sort'")
        if order != "coherent" and rays_o.dim() == 2 and rays_o.shape[0] >= 4096 and rays_o.is_cuda:
            if order == "auto" and float(self.tile_spread(rays_o, viewdirs)) > self.INCOHERENT_TILE_SPREAD:
                order = "sort"
                if not FourierGridRenderer._warned_incoherent:
                    FourierGridRenderer._warned_incoherent = True
                    import warnings
                    warnings.warn("FourierGridRenderer: the ray list is not in pixel-block order (64 consecutive rays are "
                                  "unrelated); sorting it on the device -- pass ray_order='coherent' to skip the check, "
                                  "or use render_view / pixel_tile_order", stacklevel=2)
            if order == "sort":
                perm = self.morton_ray_order(rays_o, viewdirs)
                kw = dict(render_kwargs, ray_order="coherent")
                res = self.forward(rays_o.index_select(0, perm), rays_d.index_select(0, perm), viewdirs.index_select(0, perm), **kw)
                out = dict(res)
                for k in ("rgb_marched", "depth", "alphainv_last", "wsum_mid"):
                    if k in res:
                        out[k] = torch.empty_like(res[k]).index_copy_(0, perm, res[k])
                return out
        assert rays_o.dim() == 2 and rays_o.shape[-1] == 3, "Only support point queries in [N, 3] format"
        _lib.require_cuda(("rays_o", rays_o), ("rays_d", rays_d), ("viewdirs", viewdirs))
        _lib.require_f32(("rays_o", rays_o), ("rays_d", rays_d), ("viewdirs", viewdirs))
        for name, t in (("rays_o", rays_o), ("rays_d", rays_d), ("viewdirs", viewdirs)):
            if t.device.type != "cuda" or (self.device.index is not None and t.device.index != self.device.index):
                raise RuntimeError("%s is on %s, the renderer's grids are on %s" % (name, t.device, self.device))
        stepsize = render_kwargs["stepsize"]
        t_tab, s_tab, S = self.tables(stepsize)
        R = rays_o.shape[0]
        dev = self.device
        rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
        depth = torch.empty(R, dtype=torch.float32, device=dev)
        last = torch.empty(R, dtype=torch.float32, device=dev)
        wmid = torch.empty(R, dtype=torch.float32, device=dev) if self.dc is not None else None
        dcp = None
        if self.dc is not None:
            dcp = _lib.DcvgoParams()
            dcp.mask = self.dc["mask"].data_ptr()
            dcp.mask_x, dcp.mask_y, dcp.mask_z = [int(x) for x in self.dc["mask"].shape]
            for i in range(3):
                dcp.xyz2ijk_scale[i], dcp.xyz2ijk_shift[i] = self.dc["scale"][i], self.dc["shift"][i]
            dcp.dist_thres = (2 + 2 * self.bg_len) / self.world_len * stepsize * 0.95      # dcvgo.py:285
        dvp = None
        if self.dv is not None:
            dvp = _lib.DvgoParams()
            dvp.mask = self.dv["mask"].data_ptr()
            dvp.mask_x, dvp.mask_y, dvp.mask_z = [int(x) for x in self.dv["mask"].shape]
            for i in range(3):
                dvp.xyz2ijk_scale[i], dvp.xyz2ijk_shift[i] = self.dv["scale"][i], self.dv["shift"][i]
            dvp.near_clip, dvp.far_clip = float(render_kwargs["near"]), 1e9       # dvgo.py:318: the given far is ignored
            dvp.stepdist = self.stepdist(stepsize)
        timing = render_kwargs.get("timing")  # optional list collecting ([ev0, ev1, ev2], n_rays) per launch group
        with _lib.guard(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            if self.pipeline > 1 and R >= 64 * 64 * self.pipeline and self.dc is None and self.dv is None:
                self._forward_pipelined(rays_o, rays_d, viewdirs, t_tab, s_tab, S, stepsize, last, depth, rgb, timing)
            else:
                chunk = self.rays_per_chunk(S)
                ws = self._workspace(min(R, chunk), S)
                for b in range(0, R, chunk):
                    e = min(R, b + chunk)
                    n = e - b
                    p = self._params(n, S, stepsize)
                    o_, d_, v_ = rays_o[b:e], rays_d[b:e], viewdirs[b:e]
                    if timing is not None:
                        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                        ev[0].record()
                    if dcp is not None:
                        _lib.check(_L.ugrid_render_march_dcvgo(p, ctypes.byref(dcp), _p(o_), _p(d_), _p(t_tab), _p(s_tab),
                                                               _p(self.density_bricks), _p(last[b:e]), _p(depth[b:e]), _p(wmid[b:e]),
                                                               _p(ws), st), "render_march_dcvgo")
                    elif dvp is not None:
                        _lib.check(_L.ugrid_render_march_dvgo(p, ctypes.byref(dvp), _p(o_), _p(d_), _p(self.density_bricks),
                                                              _p(last[b:e]), _p(depth[b:e]), _p(ws), st), "render_march_dvgo")
                    else:
                        _lib.check(_L.ugrid_render_march(p, _p(o_), _p(d_), _p(t_tab), _p(s_tab), _p(self.density_bricks),
                                                         _p(last[b:e]), _p(depth[b:e]), _p(ws), st), "render_march")
                    if timing is not None:
                        ev[1].record()
                    _lib.check(_L.ugrid_render_shade(p, _p(v_), _p(self.k0_bricks), _p(self.mlp_packed), _p(ws),
                                                     _p(rgb[b:e]), st), "render_shade")
                    if timing is not None:
                        ev[2].record()
                        timing.append((ev, n))
                    self._last = ("split", n, S)
        out = {"alphainv_last": last, "rgb_marched": rgb, "n_max": S}
        if self.dc is not None:
            out["wsum_mid"] = wmid
        if (self.dc is not None or self.dv is not None) and "bg" in render_kwargs:
            rgb += last.unsqueeze(-1) * render_kwargs["bg"]   # dcvgo.py:349-352 / dvgo.py:405: rgb_marched += alphainv_last * bg
        if render_kwargs.get("render_depth", False):
            out["depth"] = depth
        return out

    __call__ = forward

    def _forward_pipelined(self, rays_o, rays_d, viewdirs, t_tab, s_tab, S, stepsize, last, depth, rgb, timing):
        """Software pipeline over ray chunks on two HIP streams: k_march of chunk k+1 runs while k_shade of chunk k
        does.  The two kernels stress different units (march: VALU; shade: vector-memory latency + MFMA), so they
        co-reside on the CUs instead of running back to back.  Three work lists rotate so that a march never
        waits for the shade two chunks back.  Results are independent of the chunking (tested)."""
        dev = self.device
        R = rays_o.shape[0]
        nck = self.pipeline
        per = ((R + nck - 1) // nck + 63) // 64 * 64
        bounds = [(b, min(R, b + per)) for b in range(0, R, per)]
        need = _L.ugrid_render_ws_bytes(per, S)
        n_ws = min(3, len(bounds))
        if self._ws_ring is None or len(self._ws_ring) < n_ws or self._ws_ring[0].numel() < need:
            self._ws_ring = None
            self._ws_ring = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(n_ws)]
        if self._streams is None:
            self._streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1))
        sm, ss = self._streams
        cur = torch.cuda.current_stream(dev)
        start = cur.record_event()
        sm.wait_event(start)
        ss.wait_event(start)
        shade_done = []
        for k, (b, e) in enumerate(bounds):
            n = e - b
            p = self._params(n, S, stepsize)
            ws = self._ws_ring[k % n_ws]
            o_, d_, v_ = rays_o[b:e], rays_d[b:e], viewdirs[b:e]
            ev = [torch.cuda.Event(enable_timing=timing is not None) for _ in range(4)]
            if k >= n_ws:
                sm.wait_event(shade_done[k - n_ws])
            if timing is not None:
                ev[0].record(sm)
            _lib.check(_L.ugrid_render_march(p, _p(o_), _p(d_), _p(t_tab), _p(s_tab), _p(self.density_bricks),
                                             _p(last[b:e]), _p(depth[b:e]), _p(ws), sm.cuda_stream), "render_march")
            ev[1].record(sm)
            ss.wait_event(ev[1])
            if timing is not None:
                ev[2].record(ss)
            _lib.check(_L.ugrid_render_shade(p, _p(v_), _p(self.k0_bricks), _p(self.mlp_packed), _p(ws),
                                             _p(rgb[b:e]), ss.cuda_stream), "render_shade")
            ev[3].record(ss)
            shade_done.append(ev[3])
            if timing is not None:
                timing.append((ev, n))
        cur.wait_event(shade_done[-1])   # the shade stream is in order: the last shade implies all of them
        cur.wait_event(ev[1])
        self._ws = self._ws_ring[(len(bounds) - 1) % n_ws]
        self._last = ("split", bounds[-1][1] - bounds[-1][0], S)

    @torch.no_grad()
    def survivors_of_last_chunk(self, n_rays=None, S=None):
        """Number of surviving samples M counted by the most recent launch (the last chunk of the call).  Host sync."""
        _, n, S_ = self._last
        out = torch.zeros(1, dtype=torch.int64, device=self.device)
        with _lib.guard(self.device):
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(_L.ugrid_render_stats(_p(self._ws), n, S_, _p(out), st), "render_stats")
        return int(out.item())

    # -- constructors ------------------------------------------------------------------------------
    # -- frame-level entry point --------------------------------------------
    @torch.no_grad()
    def render_view(self, H, W, K, c2w, stepsize, inverse_y=False, flip_x=False, flip_y=False, group=None,
                    interleave=False):
        """One whole view, like the body of the reference's render_viewpoints loop (run_render.py:41-70) but
        without its 8192-ray chunking: rays are generated on the device, rendered in one fused pass and, when a
        process group is initialised, sharded over its ranks (contiguous 64-aligned ranges, or 64-ray tiles dealt
        round-robin with interleave=True) with one all-gather of the [R,5] tiles (dist.py).  The rays are rendered in
        8 x 8 pixel blocks (pixel_tile_order) when H and W allow it and the results put back in image order.
        Returns rgb [H,W,3], depth [H,W], bgmap [H,W] (= alphainv_last) on the device."""
        from .dist import render_sharded
        c2w = torch.as_tensor(c2w, dtype=torch.float32).to(self.device)
        order = pixel_tile_order(H, W, self.device) if c2w.is_cuda else None
        if order is not None:      # rays in 8 x 8 pixel blocks (one block per wave of the march kernel), results un-tiled
            ro, rd, vd = get_rays_of_pixel_index(H, W, K, c2w, order, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
        else:
            ro, rd, vd = get_rays_of_a_view(H, W, K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
            ro, rd, vd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous(), vd.reshape(-1, 3).contiguous()
        out = render_sharded(self.forward, ro, rd, vd, group=group, interleave=interleave, stepsize=stepsize,
                             ray_order="coherent" if order is not None else "auto")
        rgb, depth, last = out["rgb_marched"], out["depth"], out["alphainv_last"]
        if order is not None:
            rgb, depth, last = untile(rgb, H, W), untile(depth, H, W), untile(last, H, W)
        return rgb.reshape(H, W, 3), depth.reshape(H, W), last.reshape(H, W)

    @classmethod
    def from_reference_checkpoint(cls, ckpt, device, **kw):
        """ckpt: the dict the reference saves (`model_kwargs`, `model_state_dict`;
        FourierGrid_ckpt_manager.py:44-51), e.g. torch.load('fine_last.tar', weights_only=False).
        Models outside the fused kernels' shapes (rgbnet other than 3 x 128 wide, (F, C, viewbase_pe) triples that
        ugrid_shade.hip does not instantiate, colour grid at another resolution than the density grid) come back as a
        ComposedFourierGridRenderer: same call signature and outputs, the drop-in kernels composed instead of fused."""
        if cls is FourierGridRenderer and not fused_shape_supported(ckpt):
            return ComposedFourierGridRenderer(ckpt, device)
        return cls(state_from_reference_checkpoint(ckpt), device, **kw)


def render_view_of(render_rays, device, H, W, K, c2w, inverse_y=False, flip_x=False, flip_y=False, **render_kwargs):
    """One whole view through a per-ray renderer (`render_rays(rays_o, rays_d, viewdirs, **render_kwargs)` -> dict of per-ray
    tensors: the bounded / contracted VoxGO renderers' fused paths): rays generated on the device in 8 x 8 pixel blocks, one
    pass, results put back in image order -- the body of the reference's render loop (run_render.py:41-70) without its 8192-ray
    chunks.  Returns {key: [H,W(,3)]} on the device."""
    dev = torch.device(device)
    c2w = torch.as_tensor(c2w, dtype=torch.float32).to(dev)
    order = pixel_tile_order(H, W, dev)
    if order is not None:
        ro, rd, vd = get_rays_of_pixel_index(H, W, K, c2w, order, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
        out = render_rays(ro, rd, vd, ray_order="coherent", **render_kwargs)
        return {k: untile(v, H, W).reshape(H, W, *v.shape[1:]) for k, v in out.items() if torch.is_tensor(v)}
    ro, rd, vd = get_rays_of_a_view(H, W, K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
    out = render_rays(ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous(), vd.reshape(-1, 3).contiguous(), **render_kwargs)
    return {k: v.reshape(H, W, *v.shape[1:]) for k, v in out.items() if torch.is_tensor(v)}


# (F, C, viewbase_pe) triples instantiated by csrc/ugrid_shade.hip for rgbnet models
_FUSED_TRIPLES = None


def fused_shape_supported(ckpt):
    """Can the fused march / shade kernels render this reference checkpoint?  (rgbnet depth 3 x width <= 128 or no rgbnet,
    one grid resolution, fast_color_thres > 0, an instantiated (F, C, viewbase_pe) triple)"""
    kw, sd = ckpt['model_kwargs'], ckpt['model_state_dict']
    if kw.get('fast_color_thres', 0) <= 0 or tuple(sd['density.grid'].shape[2:]) != tuple(sd['k0.grid'].shape[2:]):
        return False
    F, C, pe = int(kw.get('fourier_freq_num', 5)), int(sd['k0.grid'].shape[1]), int(kw.get('viewbase_pe', 4))
    if kw.get('rgbnet_dim', 0) <= 0:
        return C == 3 and sd['k0.grid'].shape[0] == 1
    if kw.get('rgbnet_depth', 3) != 3 or not (1 <= kw.get('rgbnet_width', 128) <= 128):
        return False
    return bool(_L.ugrid_shade_supported(F, C, pe))


class ComposedFourierGridRenderer:
    """Inference for FourierGrid models the fused kernels do not cover: the drop-in HIP kernels composed by
    fourier_model.FourierGridModel in eval mode (TrainMarch stage 1, grid lookups on the canonical / channel-last layout,
    rgbnet through the BLAS, per-ray sums) -- any rgbnet width / depth, any (F, C, viewbase_pe), separate grid
    resolutions.  Same forward / render_view interface and output keys as FourierGridRenderer; rays are processed in
    chunks of `rays_per_chunk` like the reference's render loop (run_render.py:52-58)."""

    def __init__(self, ckpt, device, rays_per_chunk=8192):
        from .fourier_model import FourierGridModel
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("ComposedFourierGridRenderer needs a HIP device (no CPU path)")
        self.device = dev
        self.rays_per_chunk = int(rays_per_chunk)
        self.model = FourierGridModel(**dict(ckpt['model_kwargs']))
        self.model.load_state_dict(ckpt['model_state_dict'])
        self.model.to(dev).eval()

    @torch.no_grad()
    def forward(self, rays_o, rays_d, viewdirs, global_step=None, is_train=False, **render_kwargs):
        if is_train or global_step is not None:
            raise RuntimeError("inference-only; train through fourier_model.FourierGridModel")
        keys = ("rgb_marched", "alphainv_last") + (("depth",) if render_kwargs.get("render_depth", False) else ())
        kw = {k: v for k, v in render_kwargs.items() if k != "timing"}
        outs = []
        for b in range(0, rays_o.shape[0], self.rays_per_chunk):
            e = b + self.rays_per_chunk
            r = self.model(rays_o[b:e].contiguous(), rays_d[b:e].contiguous(), viewdirs[b:e].contiguous(), **kw)
            outs.append({k: r[k] for k in keys})
        if not outs:
            return {k: torch.empty((0, 3) if k == "rgb_marched" else (0,), device=self.device) for k in keys}
        return {k: torch.cat([o[k] for o in outs]) for k in keys}

    __call__ = forward
    render_view = FourierGridRenderer.render_view


def state_from_reference_checkpoint(ckpt):
    """Reference checkpoint -> the plain `state` dict of FourierGridRenderer.  Derived quantities are recomputed
    with the reference's own formulas (FourierGrid_model.py:100-122,173,335-349); note that the checkpoint's
    model_kwargs['xyz_min'/'xyz_max'] are the CONTRACTED bounds (-1-bg_len .. 1+bg_len, get_kwargs :352-353) while
    the scene box survives only as the scene_center / scene_radius buffers of the state dict."""
    mk, sd = ckpt["model_kwargs"], ckpt["model_state_dict"]
    bg_len = float(mk.get("bg_len", 0.2))
    lo = torch.Tensor([-1, -1, -1]) - bg_len
    hi = torch.Tensor([1, 1, 1]) + bg_len
    vol = (hi - lo).prod()
    vs = (vol / mk["num_voxels_density"]).pow(1 / 3)
    vs_base = (vol / mk["num_voxels_base_density"]).pow(1 / 3)
    world = ((hi - lo) / vs).long()
    ws_, bs_ = [], []
    if any(k.startswith("rgbnet.") for k in sd):
        names = sorted({k.rsplit(".", 1)[0] for k in sd if k.startswith("rgbnet.")},
                       key=lambda n: [int(x) for x in n.split(".")[1:]])
        ws_ = [sd[n + ".weight"] for n in names]
        bs_ = [sd[n + ".bias"] for n in names]
    thres = mk.get("fast_color_thres", 0)
    if isinstance(thres, dict):
        thres = thres[max(thres)]
    return {
        "density_grid": sd["density.grid"], "k0_grid": sd["k0.grid"],
        "rgbnet_weights": ws_, "rgbnet_biases": bs_,
        "scene_center": sd["scene_center"], "scene_radius": sd["scene_radius"],
        "xyz_min": sd.get("xyz_min", lo), "xyz_max": sd.get("xyz_max", hi),
        "bg_len": bg_len, "fourier_freq_num": int(mk.get("fourier_freq_num", 5)),
        "viewbase_pe": int(mk.get("viewbase_pe", 4)),
        "act_shift": float(sd["act_shift"]) if "act_shift" in sd else math.log(1 / (1 - mk["alpha_init"]) - 1),
        "voxel_size_ratio": float(vs / vs_base),
        "fast_color_thres": thres,
        "contracted_norm": mk.get("contracted_norm", "inf"),
        "world_len": int(world[0]),
    }


_TILE_ORDER = {}
RAY_TILE = 8      # a wave of the march kernel owns 64 rays: 8 x 8 pixels instead of a 64-pixel row segment


TILE_MORTON = os.environ.get("UGRID_TILE_MORTON", "1") != "0"      # lanes inside an 8 x 8 block in Z-order (below)


def _morton_ok(tile):
    return TILE_MORTON and tile == 8


def pixel_tile_order(H, W, device, tile=None):
    """Flat pixel indices j*W+i of a view in tile-major order (tile x tile pixel blocks, row-major over the blocks), int64 [H*W]
    on `device`, cached -- or None when H or W is not a multiple of the tile.  Rendering a frame's rays in this order gives every
    64-ray wave of the march kernel an 8 x 8 pixel block: its rays end at similar depths (the wave leaves the sample loop when
    its LAST ray is done) and share grid cells (5 % on the S1 frame).
    Inside an 8 x 8 block the lanes follow the Z-order curve (lane bits y2 x2 y1 x1 y0 x0; TILE_MORTON, round 4): the vector
    memory path serves a wave instruction four lanes at a time, and a 2 x 2 pixel quad spans fewer grid cells -- fewer tag
    look-ups per instruction -- than four pixels of a row (profiles/r04/tile_morton_ab.txt).  Other tile sizes stay row-major
    inside the block.  Per-ray results do not depend on the order."""
    tile = RAY_TILE if tile is None else int(tile)
    if tile <= 1 or H % tile or W % tile:
        return None
    key = (H, W, tile, str(device), _morton_ok(tile))
    p = _TILE_ORDER.get(key)
    if p is None:
        a = torch.arange(H * W, device=device)
        if _morton_ok(tile):
            p = a.view(H // 8, 2, 2, 2, W // 8, 2, 2, 2).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(-1).contiguous()
        else:
            p = a.view(H // tile, tile, W // tile, tile).permute(0, 2, 1, 3).reshape(-1).contiguous()
        if len(_TILE_ORDER) > 8:
            _TILE_ORDER.clear()
        if p.is_cuda and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(p.device).synchronize()      # cached for EVERY stream (run_render keeps two views in flight): built once, complete
        _TILE_ORDER[key] = p
    return p


def untile(x, H, W, tile=None):
    """per-ray results [H*W, ...] in pixel_tile_order -> image order [H*W, ...]"""
    tile = RAY_TILE if tile is None else int(tile)
    rest = x.shape[1:]
    n = len(rest)
    if _morton_ok(tile):      # [H/8, W/8, y2, x2, y1, x1, y0, x0, ...] -> [H/8, y2, y1, y0, W/8, x2, x1, x0, ...]
        return x.view(H // 8, W // 8, 2, 2, 2, 2, 2, 2, *rest).permute(0, 2, 4, 6, 1, 3, 5, 7, *range(8, 8 + n)).reshape(H * W, *rest)
    return x.view(H // tile, W // tile, tile, tile, *rest).permute(0, 2, 1, 3, *range(4, 4 + n)).reshape(H * W, *rest)


def pixel_grid(H, W, device, flip_x=False, flip_y=False, mode="center"):
    """Pixel-centre coordinates (ii = column, jj = row) of a view, [H,W] each -- the first half of get_rays_of_a_view,
    split out so that a rank of a sharded render can keep the coordinates of ITS pixels and generate only those rays."""
    jj, ii = torch.meshgrid(torch.linspace(0, H - 1, H, device=device), torch.linspace(0, W - 1, W, device=device),
                            indexing="ij")
    if mode == "center":
        ii, jj = ii + 0.5, jj + 0.5
    elif mode != "lefttop":
        raise NotImplementedError(mode)
    if flip_x:
        ii = ii.flip((1,))
    if flip_y:
        jj = jj.flip((0,))
    return ii, jj


def get_rays_of_pixels(ii, jj, K, c2w, inverse_y=False):
    """Rays of the given pixel coordinates (any shape [...]): same arithmetic as get_rays_of_a_view, so a sharded frame
    is bit-identical to the whole frame."""
    K = torch.as_tensor(K, dtype=torch.float32, device=c2w.device)
    if inverse_y:
        dirs = torch.stack([(ii - K[0][2]) / K[0][0], (jj - K[1][2]) / K[1][1], torch.ones_like(ii)], -1)
    else:
        dirs = torch.stack([(ii - K[0][2]) / K[0][0], -(jj - K[1][2]) / K[1][1], -torch.ones_like(ii)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    return rays_o.contiguous(), rays_d.contiguous(), viewdirs.contiguous()


def _rays_native(H, W, K, c2w, inverse_y, flip_x, flip_y, mode, pixel_index=None):
    """One-kernel ray generation (ugrid_rays_of_a_view) for a device-resident camera pose."""
    if mode not in ("center", "lefttop"):
        raise NotImplementedError(mode)
    K9 = (ctypes.c_float * 9)(*[float(x) for x in (K.reshape(-1).tolist() if torch.is_tensor(K) else
                                                   [v for row in K for v in row])])
    c2w = c2w[:3, :4].to(torch.float32).contiguous()
    dev = c2w.device
    n = H * W if pixel_index is None else int(pixel_index.numel())
    o = torch.empty(n, 3, dtype=torch.float32, device=dev)
    d = torch.empty(n, 3, dtype=torch.float32, device=dev)
    v = torch.empty(n, 3, dtype=torch.float32, device=dev)
    with _lib.guard(dev):
        _lib.check(_L.ugrid_rays_of_a_view(H, W, ctypes.cast(K9, ctypes.c_void_p), _p(c2w), int(bool(inverse_y)), int(bool(flip_x)),
                                           int(bool(flip_y)), int(mode == "center"), _p(pixel_index), n, _p(o), _p(d), _p(v),
                                           torch.cuda.current_stream(dev).cuda_stream), "rays_of_a_view")
    return o, d, v


def get_rays_of_pixel_index(H, W, K, c2w, pixel_index, inverse_y=False, flip_x=False, flip_y=False, mode="center"):
    """Rays of the listed flat pixel indices j*W+i (int64, on c2w's device) of a view: [n,3] each -- a rank's shard of a
    frame in one launch; bit-identical to the corresponding rows of get_rays_of_a_view."""
    if not c2w.is_cuda:
        o, d, v = get_rays_of_a_view(H, W, K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y, mode=mode)
        return o.reshape(-1, 3)[pixel_index], d.reshape(-1, 3)[pixel_index], v.reshape(-1, 3)[pixel_index]
    return _rays_native(H, W, K, c2w, inverse_y, flip_x, flip_y, mode, pixel_index.contiguous())


def get_rays_of_a_view(H, W, K, c2w, inverse_y=False, flip_x=False, flip_y=False, mode="center"):
    """Pinhole rays of one view, pixel centres (+0.5): rays_o, rays_d, viewdirs, each [H,W,3], on c2w's
    device.  Same conventions as the reference (dvgo.py:493-521,554-559; no NDC).  Device-resident pose: one HIP
    kernel; host tensors: the torch elementwise chain (host-side utility for tests and data preparation)."""
    if c2w.is_cuda:
        o, d, v = _rays_native(H, W, K, c2w, inverse_y, flip_x, flip_y, mode)
        return o.view(H, W, 3), d.view(H, W, 3), v.view(H, W, 3)
    dev = c2w.device
    K = torch.as_tensor(K, dtype=torch.float32, device=dev)
    jj, ii = torch.meshgrid(torch.linspace(0, H - 1, H, device=dev), torch.linspace(0, W - 1, W, device=dev),
                            indexing="ij")
    if mode == "center":
        ii, jj = ii + 0.5, jj + 0.5
    elif mode != "lefttop":
        raise NotImplementedError(mode)
    if flip_x:
        ii = ii.flip((1,))
    if flip_y:
        jj = jj.flip((0,))
    if inverse_y:
        dirs = torch.stack([(ii - K[0][2]) / K[0][0], (jj - K[1][2]) / K[1][1], torch.ones_like(ii)], -1)
    else:
        dirs = torch.stack([(ii - K[0][2]) / K[0][0], -(jj - K[1][2]) / K[1][1], -torch.ones_like(ii)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    return rays_o.contiguous(), rays_d.contiguous(), viewdirs.contiguous()
