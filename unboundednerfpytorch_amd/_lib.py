"""ctypes binding of libugrid_hip.so (C ABI declared in include/ugrid_hip.h).

The shared library is the product: if it is missing or fails to load, importing any op module of
this package raises -- there is NO CPU / eager fallback (oracle/ is test infrastructure and is never
imported from here).

torch is imported first on purpose: libugrid_hip.so needs `libamdhip64.so.7`; PyTorch-ROCm ships its
own copy with that SONAME, and loading it first makes the dynamic loader bind our library to the
SAME HIP runtime instance torch uses, so torch's device pointers and streams are valid in our launches.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UGRID_LIB") or os.path.join(_HERE, "libugrid_hip.so")  # UGRID_LIB: A/B builds only
ABI_VERSION = 2

_c = ctypes
_P = _c.c_void_p
_F = _c.c_float
_I = _c.c_int
_L = _c.c_int64


MLP_FP32, MLP_BF16X3, MLP_FP16X2 = 0, 1, 2   # ugrid_render_params.mlp_mode (include/ugrid_hip.h)
MLP_RESIDUAL = 0x100                         # flag: rgb = sigmoid(rgbnet([k0[3:], emb]) + k0[:3])  (DirectVoxGO, rgbnet_direct = False)


class RenderParams(_c.Structure):
    """Mirror of `ugrid_render_params` (include/ugrid_hip.h)."""
    _fields_ = [
        ("n_rays", _c.c_int64),
        ("n_samples", _c.c_int32), ("freq_num", _c.c_int32),
        ("grid_x", _c.c_int32), ("grid_y", _c.c_int32), ("grid_z", _c.c_int32),
        ("k0_channels", _c.c_int32), ("mlp_in", _c.c_int32), ("mlp_width", _c.c_int32),
        ("viewbase_pe", _c.c_int32), ("norm_l2", _c.c_int32),
        ("scene_center", _c.c_float * 3), ("scene_radius", _c.c_float * 3),
        ("xyz_min", _c.c_float * 3), ("xyz_max", _c.c_float * 3),
        ("bg_len", _c.c_double),
        ("act_shift", _c.c_float), ("interval", _c.c_float), ("thres", _c.c_float),
        ("mlp_mode", _c.c_int32),
    ]


class DcvgoParams(_c.Structure):
    """Mirror of `ugrid_dcvgo_params` (include/ugrid_hip.h)."""
    _fields_ = [("mask", _c.c_void_p), ("mask_x", _c.c_int32), ("mask_y", _c.c_int32), ("mask_z", _c.c_int32),
                ("xyz2ijk_scale", _c.c_float * 3), ("xyz2ijk_shift", _c.c_float * 3), ("dist_thres", _c.c_float)]


class DvgoParams(_c.Structure):
    """Mirror of `ugrid_dvgo_params` (include/ugrid_hip.h)."""
    _fields_ = [("mask", _c.c_void_p), ("mask_x", _c.c_int32), ("mask_y", _c.c_int32), ("mask_z", _c.c_int32),
                ("xyz2ijk_scale", _c.c_float * 3), ("xyz2ijk_shift", _c.c_float * 3), ("near_clip", _c.c_float),
                ("far_clip", _c.c_float), ("stepdist", _c.c_float)]


class AdamItem(_c.Structure):
    """ugrid_adam_item (include/ugrid_hip.h)"""
    _fields_ = [("param", _c.c_void_p), ("grad", _c.c_void_p), ("exp_avg", _c.c_void_p), ("exp_avg_sq", _c.c_void_p),
                ("numel", _c.c_int64), ("step", _c.c_int32), ("lr", _c.c_float)]


class VoxgoStep(_c.Structure):
    """Mirror of `ugrid_voxgo_step` (include/ugrid_hip.h); tests/test_capi.py compares the field list with the header's"""
    _fields_ = [
        ("mode", _c.c_int32), ("k0_channels_last", _c.c_int32),
        ("P", _c.c_int32), ("freq_num", _c.c_int32), ("kP", _c.c_int32), ("k0_freq_num", _c.c_int32),
        ("X", _c.c_int32), ("Y", _c.c_int32), ("Z", _c.c_int32),
        ("kX", _c.c_int32), ("kY", _c.c_int32), ("kZ", _c.c_int32), ("C", _c.c_int32),
        ("pe", _c.c_int32), ("width", _c.c_int32), ("slots", _c.c_int32), ("norm_l2", _c.c_int32),
        ("mask_dims", _c.c_int32 * 3),
        ("mask_scale", _c.c_float * 3), ("mask_shift", _c.c_float * 3), ("scene_center", _c.c_float * 3), ("scene_radius", _c.c_float * 3),
        ("act_shift", _c.c_float), ("interval", _c.c_float), ("thres", _c.c_float), ("near_clip", _c.c_float), ("far_clip", _c.c_float),
        ("stepdist", _c.c_float), ("dist_thres", _c.c_float),
        ("coef9", _c.c_float * 9),
        ("bg_len", _c.c_double),
        ("n_rays", _c.c_int64),
        ("density_grid", _P), ("k0_grid", _P), ("xyz_min", _P), ("xyz_max", _P), ("k0_xyz_min", _P), ("k0_xyz_max", _P),
        ("mask", _P), ("t_table", _P), ("viewfreq", _P),
        ("w0", _P), ("b0", _P), ("w1", _P), ("b1", _P), ("w2", _P), ("b2", _P),
        ("rays_o", _P), ("rays_d", _P), ("viewdirs", _P), ("target", _P), ("bg", _P),
        ("sc_pts", _P), ("sc_density", _P), ("sc_step", _P), ("sc_w", _P), ("sc_T", _P),
        ("counts", _P), ("offsets", _P), ("totals", _P), ("alphainv_last", _P), ("seg", _P),
        ("rgb_marched", _P), ("ray_tot", _P), ("partial", _P), ("out2", _P),
        ("M1", _c.c_int64), ("M2", _c.c_int64), ("hint1", _c.c_int64), ("hint2", _c.c_int64), ("sync_free", _c.c_int32), ("reserved_", _c.c_int32),
        ("ws", _P),
        ("density2", _P), ("alpha2", _P), ("weights2", _P), ("t2", _P), ("ray_id2", _P), ("step_id2", _P), ("inner2", _P), ("logits", _P),
        ("grad_loss", _P), ("ws_bwd", _P),
        ("g_w0", _P), ("g_b0", _P), ("g_w1", _P), ("g_b1", _P), ("g_w2", _P), ("g_b2", _P),
        ("grad_density_grid", _P), ("grad_k0_grid", _P), ("touch", _P),
    ]


# name -> (restype, argtypes); every int-returning entry point returns a hipError_t
_SIGNATURES = {
    "ugrid_abi_version": (_I, []),
    "ugrid_target_arch": (_c.c_char_p, []),
    "ugrid_infer_t_minmax": (_I, [_P, _P, _P, _P, _F, _F, _L, _P, _P, _P]),
    "ugrid_infer_n_samples": (_I, [_P, _P, _P, _F, _L, _P, _P]),
    "ugrid_infer_ray_start_dir": (_I, [_P, _P, _P, _L, _P, _P, _P]),
    "ugrid_scan_ws_bytes": (_L, [_L]),
    "ugrid_sample_pts_on_rays_count": (_I, [_P, _P, _P, _P, _F, _F, _F, _L, _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_sample_pts_on_rays_fill": (_I, [_P, _P, _P, _P, _P, _P, _F, _L, _L, _P, _P, _P, _P, _P]),
    "ugrid_sample_ndc_pts_on_rays": (_I, [_P, _P, _P, _P, _L, _L, _P, _P, _P]),
    "ugrid_sample_bg_pts_on_rays": (_I, [_P, _P, _P, _F, _L, _L, _P, _P]),
    "ugrid_maskcache_lookup": (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _P, _P]),
    "ugrid_raw2alpha": (_I, [_P, _F, _F, _P, _L, _P, _P, _P]),
    "ugrid_raw2alpha_backward": (_I, [_P, _P, _F, _P, _L, _P, _P]),
    "ugrid_alpha2weight": (_I, [_P, _P, _L, _L, _P, _P, _P, _P, _P, _P]),
    "ugrid_alpha2weight_backward": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _P]),
    "ugrid_total_variation_add_grad": (_I, [_P, _P, _F, _F, _F, _I, _L, _L, _L, _L, _P]),
    "ugrid_cumdist_thres": (_I, [_P, _F, _L, _L, _P, _P]),
    "ugrid_segment_cumsum": (_I, [_P, _P, _P, _L, _L, _P, _P, _P, _P, _P, _P]),
    "ugrid_adam_upd": (_I, [_P, _P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _I, _P]),
    "ugrid_adam_upd_multi": (_I, [_P, _c.c_int32, _F, _F, _F, _c.c_int32, _P]),
    "ugrid_tv_adam_dense": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _L, _L, _L, _L, _I, _F, _F, _F, _F, _I, _P]),
    "ugrid_grid_query": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _L, _P, _P]),
    "ugrid_grid_query_backward": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _L, _P, _P]),
    "ugrid_rays_of_a_view": (_I, [_c.c_int32, _c.c_int32, _P, _P, _I, _I, _I, _I, _P, _L, _P, _P, _P, _P]),
    "ugrid_train_march": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _L, _P, _c.c_int32, _P, _P, _P, _P, _c.c_double, _I, _F, _F, _F,
                                _P, _P, _P, _P, _P]),
    "ugrid_train_compact": (_I, [_L, _c.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_train_sample": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _L, _P, _c.c_int32, _P, _P, _P, _P, _c.c_double, _I, _F, _F, _F,
                                 _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_train_sample_compact": (_I, [_L, _c.c_int32, _F, _F, _F] + [_P] * 23),
    "ugrid_train_sample_backward": (_I, [_L, _F, _F] + [_P] * 12),
    "ugrid_train_sample_dcvgo": (_I, [_P, _I, _I, _I, _P, _P, _L, _P, _c.c_int32, _P, _P, _P, _P, _c.c_double, _I, _F, _P, _P, _P, _P,
                                       _F, _F, _F] + [_P] * 9),
    "ugrid_train_sample_dvgo": (_I, [_P, _I, _I, _I, _P, _P, _L, _c.c_int32, _P, _P, _F, _F, _F, _P, _P, _P, _P, _F, _F, _F] + [_P] * 9),
    "ugrid_train_sample_compact_vox": (_I, [_L, _c.c_int32, _F, _F, _F] + [_P] * 24),
    "ugrid_grid_query_cl": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _L, _P, _P]),
    "ugrid_grid_query_backward_cl": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _L, _P, _P]),
    "ugrid_total_variation_add_grad_cl": (_I, [_P, _P, _F, _F, _F, _I, _L, _L, _L, _L, _L, _P]),
    "ugrid_tv_adam_dense_cl": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _L, _L, _L, _L, _L, _I, _F, _F, _F, _F, _I, _P]),
    "ugrid_rgbnet_train_scratch_floats": (_L, [_L]),
    "ugrid_rgbnet_train_forward": (_I, [_P, _L, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "ugrid_rgbnet_train_backward": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_touch_words": (_L, [_L]),
    "ugrid_grid_query_backward_cl_touch": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _L, _P, _P, _P]),
    "ugrid_total_variation_add_grad_cl_touch": (_I, [_P, _P, _F, _F, _F, _L, _L, _L, _L, _L, _P, _P]),
    "ugrid_masked_adam_upd_touch": (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _P, _P]),
    "ugrid_tv_adam_dense_cl_touch": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _L, _L, _L, _L, _L, _I, _F, _F, _F, _F, _I, _P, _P]),
    "ugrid_rgbnet_features": (_I, [_P, _I, _P, _L, _P, _I, _P, _L, _P, _P, _P]),
    "ugrid_voxgo_step_sizeof": (_L, []),
    "ugrid_voxgo_step_ws_floats": (_L, [_P]),
    "ugrid_voxgo_step_bwd_ws_floats": (_L, [_P]),
    "ugrid_voxgo_step_sample": (_I, [_P, _P]),
    "ugrid_voxgo_step_forward": (_I, [_P, _P]),
    "ugrid_voxgo_step_backward": (_I, [_P, _P]),
    "ugrid_voxgo_step_backward_k0": (_I, [_P, _P]),
    "ugrid_voxgo_step_backward_density": (_I, [_P, _P]),
    "ugrid_render_loss": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_render_loss_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_brick_bytes": (_L, [_I, _I, _I, _I, _I, _I]),
    "ugrid_pack_bricks": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "ugrid_render_ws_bytes": (_L, [_L, _c.c_int32]),
    "ugrid_render_march": (_I, [_c.POINTER(RenderParams), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_render_march_dcvgo": (_I, [_c.POINTER(RenderParams), _c.POINTER(DcvgoParams), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_render_march_dvgo": (_I, [_c.POINTER(RenderParams), _c.POINTER(DvgoParams), _P, _P, _P, _P, _P, _P, _P]),
    "ugrid_render_shade": (_I, [_c.POINTER(RenderParams), _P, _P, _P, _P, _P, _P]),
    "ugrid_mlp_packed_bytes": (_L, [_c.c_int32, _c.c_int32]),
    "ugrid_pack_mlp": (_I, [_P, _P, _P, _P, _P, _P, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_float, _P,
                            _c.POINTER(_c.c_int32), _P]),
    "ugrid_mlp_fp16x2_scales": (_I, [_P, _P, _P, _c.c_int32, _c.c_int32, _c.c_float, _P]),
    "ugrid_tune": (_I, [_c.c_char_p, _I]),
    "ugrid_shade_supported": (_I, [_c.c_int32, _c.c_int32, _c.c_int32]),
    "ugrid_render_stats": (_I, [_P, _L, _c.c_int32, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# the entry points the reference dispatches on the tensor type (AT_DISPATCH_FLOATING_TYPES): their double instantiations live in a
# library of their own, include/ugrid_hip_f64.h -> libugrid_hip_f64.so (same signatures, array pointers double; the scan of
# sample_pts_on_rays_count needs no workspace there)
LIB_F64_PATH = os.path.join(_HERE, "libugrid_hip_f64.so")
_F64_TWINS = ("ugrid_infer_t_minmax", "ugrid_infer_n_samples", "ugrid_infer_ray_start_dir", "ugrid_sample_pts_on_rays_count",
              "ugrid_sample_pts_on_rays_fill", "ugrid_sample_ndc_pts_on_rays", "ugrid_sample_bg_pts_on_rays", "ugrid_maskcache_lookup",
              "ugrid_raw2alpha", "ugrid_raw2alpha_backward", "ugrid_alpha2weight", "ugrid_alpha2weight_backward",
              "ugrid_total_variation_add_grad", "ugrid_cumdist_thres", "ugrid_adam_upd")
_SIGNATURES_F64 = {n + "_f64": _SIGNATURES[n] for n in _F64_TWINS}
_SIGNATURES_F64["ugrid_sample_pts_on_rays_count_f64"] = (_I, [_P, _P, _P, _P, _F, _F, _F, _L, _P, _P, _P, _P, _P, _P])
EXPORTED_SYMBOLS_F64 = tuple(_SIGNATURES_F64)

_lib = None
_lib_f64 = None


def load_f64():
    """The fp64 twins (loaded on first use: nothing on the rendering / training path needs them).  Raises if the library is absent."""
    global _lib_f64
    if _lib_f64 is not None:
        return _lib_f64
    if not os.path.exists(LIB_F64_PATH):
        raise ImportError("libugrid_hip_f64.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'`.  "
                          "There is no CPU fallback." % LIB_F64_PATH)
    lib = ctypes.CDLL(LIB_F64_PATH)
    for name, (res, args) in _SIGNATURES_F64.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib_f64 = lib
    return lib


def real_dtype(*named):
    """The common floating type of the tensors of a reference-dispatched op: torch.float32 or torch.float64 (the two types of the
    reference's AT_DISPATCH_FLOATING_TYPES).  RuntimeError for anything else, or for a mix (the reference reinterprets every array
    with the first tensor's type: a mix is a caller's bug there too)."""
    dt = named[0][1].dtype
    if dt not in (torch.float32, torch.float64):
        raise RuntimeError("%s: float32 or float64 expected (got %s)" % (named[0][0], dt))
    for name, t in named[1:]:
        if t.dtype != dt:
            raise RuntimeError("%s is %s, %s is %s: one floating type per call" % (named[0][0], dt, name, t.dtype))
    return dt


def entry(name, dtype):
    """the C entry point `name` for tensors of `dtype`: libugrid_hip.so's, or its `_f64` twin of libugrid_hip_f64.so"""
    if dtype == torch.float64:
        return getattr(load_f64(), name + "_f64")
    return getattr(load(), name)


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libugrid_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or unboundednerfpytorch_amd/csrc/build.sh.  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.ugrid_abi_version() != ABI_VERSION:
        raise ImportError("libugrid_hip.so ABI %d != expected %d" % (lib.ugrid_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def guard(device):
    """Device guard for a launch on `device` (a torch.device or something torch.device() accepts): the reference's
    extensions have none (render_utils.cpp launches on whatever device is current); ours must select the tensors'
    device for multi-GPU use, but switching contexts costs several microseconds per call, so it is a no-op when that
    device is already current -- the common case, one process per GPU."""
    dev = device if isinstance(device, torch.device) else torch.device(device)
    if dev.index is None or dev.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def check(err, what):
    if err != 0:
        raise RuntimeError("%s failed: hipError_t %d" % (what, err))


def ptr(t):
    return t.data_ptr() if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t):
    """Current torch stream of t's device as a hipStream_t value.  (torch's raw accessor where it exists: building a
    torch.cuda.Stream object per launch cost ~6 us, 0.07 ms of a 1.3 ms training step)"""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def require_cuda(*named):
    """Mirror of the reference's CHECK_INPUT (render_utils.cpp:46-48): RuntimeError on host or
    non-contiguous tensors, with the same wording."""
    for name, t in named:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % name)


def is_channels_last(t):
    """A 5-D [P,C,X,Y,Z] tensor stored as [P][X][Y][Z][C] (torch.channels_last_3d) -- the training layout of
    multi-channel grids (C > 1; with C == 1 the two layouts coincide and the tensor counts as canonical)."""
    return t.dim() == 5 and t.shape[1] > 1 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last_3d)


def require_cuda_grid(*named):
    """CHECK_INPUT for voxel-grid tensors: device-resident and dense in the canonical OR the channel-last layout.
    Returns True when they are channel-last (all of them must then be)."""
    cl = [is_channels_last(t) for _, t in named]
    for (name, t), c in zip(named, cl):
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)
        if not (c or t.is_contiguous()):
            raise RuntimeError("%s must be contiguous" % name)
    if any(cl) and not all(cl):
        raise RuntimeError("grid tensors %s mix the canonical and the channel-last layout" % ", ".join(n for n, _ in named))
    return all(cl) and len(cl) > 0


def wait_pending(param):
    """A grid parameter whose optimizer update was queued on a side stream (ShardedMaskedAdam.step(overlap=...)) carries
    the completion event in `_ug_pending`: make the current stream wait for it before the parameter is touched."""
    ev = getattr(param, '_ug_pending', None)
    if ev is not None:
        torch.cuda.current_stream(param.device).wait_event(ev)
        param._ug_pending = None


def empty_like_grid(shape, channels_last, device, zero=False):
    fmt = torch.channels_last_3d if channels_last else torch.contiguous_format
    t = torch.empty(tuple(shape), dtype=torch.float32, device=device, memory_format=fmt)
    return t.zero_() if zero else t


def require_f32(*named):
    for name, t in named:
        if t.dtype != torch.float32:
            raise RuntimeError("%s: float32 expected (got %s) -- this op has no reference counterpart and exists in fp32 only; the ops "
                               "the reference dispatches on the tensor type also take float64" % (name, t.dtype))
