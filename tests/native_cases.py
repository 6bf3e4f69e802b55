"""One table of seeded cases covering all 18 functions the reference's four native modules export
(render_utils.cpp:170-184, total_variation.cpp:23, ub360_utils.cpp:21, adam_upd.cpp:79-86).

Used three ways, always with identical inputs (regenerated from seeds through tests/synth.py, never stored):
  * tests/golden/gen_native_golden.py  runs the reference's OWN kernels (oracle/_ref, built by oracle/build_ref.py from
    FourierGrid/cuda/*.cu) on an MI355X and freezes their outputs in tests/golden/native_ops.npz;
  * tests/test_oracle_golden.py        pins oracle/ref_ops.c on that file (CPU);
  * tests/test_gpu_ref_native.py       pins the HIP library on it, and -- when oracle/_ref is present on the GPU box --
                                       compares live against the reference kernels at larger sizes (scale > 1).

A case is (module, function, make(scale, prev) -> args, mutated) where `prev` holds the outputs of earlier cases
(op chains such as infer_t_minmax -> infer_n_samples take the recorded outputs as inputs) and `mutated` lists the
argument positions the op updates in place (their final values are the op's result).
"""
import numpy as np
import torch

import synth


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _rays(seed, R):
    o = _t(synth.normal(seed, R * 3, 0.0, 1.5).reshape(R, 3))
    d = _t(synth.normal(seed + 1, R * 3).reshape(R, 3))
    if R >= 8:                       # axis-parallel / signed-zero directions: the slab test's inf and NaN handling
        d[1, 0] = 0.0
        d[2] = torch.tensor([0.0, 0.0, 1.0])
        d[3, 2] = -0.0
    lo = torch.tensor([-1.0, -0.9, -1.1])
    hi = torch.tensor([1.0, 1.2, 0.8])
    return o, d, lo, hi


NEAR, FAR, STEPDIST = 0.2, 1e9, 0.0173


def _infer_t_minmax(scale, prev):
    o, d, lo, hi = _rays(101, 257 * scale)
    return [o, d, lo, hi, NEAR, FAR]


def _infer_n_samples(scale, prev):
    o, d, lo, hi = _rays(101, 257 * scale)
    return [d, prev["infer_t_minmax"][0], prev["infer_t_minmax"][1], STEPDIST]


def _infer_ray_start_dir(scale, prev):
    o, d, lo, hi = _rays(101, 257 * scale)
    return [o, d, prev["infer_t_minmax"][0]]


def _sample_pts_on_rays(scale, prev):
    o, d, lo, hi = _rays(101, 257 * scale)
    return [o, d, lo, hi, NEAR, FAR, STEPDIST]


def _sample_ndc(scale, prev):
    o, d, lo, hi = _rays(11, 129 * scale)
    return [o, d, lo, hi, 37]


def _sample_bg(scale, prev):
    R = 129 * scale
    o, d, lo, hi = _rays(11, R)
    return [o, d, _t(synth.uniform(12, R, 0.5, 3.0)), 0.3, 37]


def _maskcache(scale, prev):
    world = _t(synth.uniform(20, 9 * 7 * 5) > 0.5).reshape(9, 7, 5)
    n = 5000 * scale
    xyz = _t(synth.uniform(21, n * 3, -1.6, 1.6).reshape(n, 3))
    xyz[:4] = torch.tensor([[-1.2, -1.2, -1.2], [1.2, 1.2, 1.2], [float("nan"), 0, 0], [1e30, 0, 0]])
    lo, hi = torch.full((3,), -1.2), torch.full((3,), 1.2)
    sc = (torch.tensor([9.0, 7.0, 5.0]) - 1) / (hi - lo)
    sh = -lo * sc
    xyz[4:40, 0] = ((torch.arange(36) * 0.5 - 2.0) - sh[0]) / sc[0]      # exact .5 positions: round half away from zero
    return [world, xyz, sc, sh]


SHIFT, INTERVAL = -9.21024, 0.5


def _dens(scale):
    n = 20_003 * scale
    dens = _t(synth.normal(30, n, 4.0, 8.0))
    dens[:3] = torch.tensor([200.0, -200.0, 0.0])
    return dens, n


def _raw2alpha(scale, prev):
    return [_dens(scale)[0], SHIFT, INTERVAL]


def _raw2alpha_bwd(scale, prev):
    dens, n = _dens(scale)
    return [prev["raw2alpha"][0], _t(synth.normal(31, n)), INTERVAL]


def _raw2alpha_nonuni(scale, prev):
    dens, n = _dens(scale)
    return [dens, SHIFT, _t(synth.uniform(32, n, 0.05, 1.5))]


def _raw2alpha_nonuni_bwd(scale, prev):
    dens, n = _dens(scale)
    return [prev["raw2alpha_nonuni"][0], _t(synth.normal(31, n)), _t(synth.uniform(32, n, 0.05, 1.5))]


A2W_N, A2W_R = 20_000, 137


def _a2w_inputs(scale):
    n, R = A2W_N * scale, A2W_R * scale
    alpha = _t(synth.uniform(40, n, 0.0, 0.3))
    alpha[n // 2:] *= 0.02                                   # half of the rays never reach the T < 1e-3 stop
    rid = np.sort((synth.uniform(41, n) * R).astype(np.int64))
    rid[rid == 3] = 4                                        # an empty ray
    return alpha, _t(np.sort(rid)), R


def _alpha2weight(scale, prev):
    alpha, rid, R = _a2w_inputs(scale)
    return [alpha, rid, R]


def _alpha2weight_bwd(scale, prev):
    alpha, rid, R = _a2w_inputs(scale)
    w, T, last, i_s, i_e = prev["alpha2weight"]
    return [alpha, w, T, last, i_s, i_e, R, _t(synth.normal(42, alpha.numel())), _t(synth.normal(43, R))]


def _tv(dense, shape):
    def make(scale, prev):
        shp = shape if scale == 1 else (shape[0], shape[1], shape[2] * 2, shape[3] * 2, shape[4] * scale)
        n = int(np.prod(shp))
        prm = _t(synth.normal(60, n, 0.0, 2.0).reshape(shp))
        g = synth.normal(61, n).reshape(shp)
        g[np.abs(g) < 0.7] = 0
        return [prm, _t(g), 0.3, 0.2, 0.1, dense]
    return make


def _cumdist(scale, prev):
    R, K = 61 * scale, 667
    return [_t(synth.uniform(70, R * K, 0.0, 0.02).reshape(R, K)), 0.0114]


def _adam(perlr):
    def make(scale, prev):
        n = 4099 * scale
        p = _t(synth.normal(80, n))
        g = synth.normal(81, n)
        g[np.abs(g) < 1.0] = 0
        m = _t(synth.normal(82, n, 0, 0.1))
        v = _t(synth.uniform(83, n, 0, 0.01))
        args = [p, _t(g), m, v]
        if perlr:
            args.append(_t(synth.uniform(84, n)))
        return args + [3, 0.9, 0.99, 0.1, 1e-8]
    return make


RU, TV, UB, AD = "render_utils_cuda", "total_variation_cuda", "ub360_utils_cuda", "adam_upd_cuda"

# name, module, function, make, mutated argument positions
CASES = [
    ("infer_t_minmax", RU, "infer_t_minmax", _infer_t_minmax, ()),
    ("infer_n_samples", RU, "infer_n_samples", _infer_n_samples, ()),
    ("infer_ray_start_dir", RU, "infer_ray_start_dir", _infer_ray_start_dir, ()),
    ("sample_pts_on_rays", RU, "sample_pts_on_rays", _sample_pts_on_rays, ()),
    ("sample_ndc_pts_on_rays", RU, "sample_ndc_pts_on_rays", _sample_ndc, ()),
    ("sample_bg_pts_on_rays", RU, "sample_bg_pts_on_rays", _sample_bg, ()),
    ("maskcache_lookup", RU, "maskcache_lookup", _maskcache, ()),
    ("raw2alpha", RU, "raw2alpha", _raw2alpha, ()),
    ("raw2alpha_backward", RU, "raw2alpha_backward", _raw2alpha_bwd, ()),
    ("raw2alpha_nonuni", RU, "raw2alpha_nonuni", _raw2alpha_nonuni, ()),
    ("raw2alpha_nonuni_backward", RU, "raw2alpha_nonuni_backward", _raw2alpha_nonuni_bwd, ()),
    ("alpha2weight", RU, "alpha2weight", _alpha2weight, ()),
    ("alpha2weight_backward", RU, "alpha2weight_backward", _alpha2weight_bwd, ()),
    ("tv_dense", TV, "total_variation_add_grad", _tv(True, (3, 2, 9, 6, 11)), (1,)),
    ("tv_masked", TV, "total_variation_add_grad", _tv(False, (2, 3, 5, 7, 12)), (1,)),
    ("cumdist_thres", UB, "cumdist_thres", _cumdist, ()),
    ("adam_upd", AD, "adam_upd", _adam(False), (0, 2, 3)),
    ("masked_adam_upd", AD, "masked_adam_upd", _adam(False), (0, 2, 3)),
    ("adam_upd_with_perlr", AD, "adam_upd_with_perlr", _adam(True), (0, 2, 3)),
]
EXPORTED = {RU: 13, TV: 1, UB: 1, AD: 3}   # m.def counts of the four reference modules (18 in total)

# outputs that go through exp / pow / log of the device libm (ocml) in the reference binary: compared in ulps /
# absolute tolerance against glibc-based or hand-written evaluations; everything else is IEEE-exact arithmetic
TRANSCENDENTAL = {"raw2alpha", "raw2alpha_backward", "raw2alpha_nonuni", "raw2alpha_nonuni_backward"}


def run_case(mods, case, scale, prev, device=None, dtype=None):
    """Call one case on `mods` (dict module name -> module with the reference's function signatures).  Returns the list
    of result tensors on the CPU: the op's return value(s), or the mutated arguments for in-place ops."""
    name, mod, fn, make, mutated = case
    args = make(scale, prev)
    if dtype is not None:       # the reference dispatches float and double: the same seeded values in the other floating type
        args = [a.to(dtype) if (torch.is_tensor(a) and a.is_floating_point()) else a for a in args]
    if device is not None:
        args = [a.to(device) if torch.is_tensor(a) else a for a in args]
    ret = getattr(mods[mod], fn)(*args)
    if mutated:
        outs = [args[i] for i in mutated]
    elif isinstance(ret, (list, tuple)):
        outs = list(ret)
    else:
        outs = [ret]
    return [o.detach().cpu() for o in outs]


def run_all(mods, scale=1, device=None, chain_from=None, dtype=None):
    """All cases in table order.  chain_from: dict name -> outputs to feed dependent cases from (the golden file), so
    that every implementation sees the same inputs; default: its own outputs.  dtype: cast the floating inputs (float64:
    the reference's other dispatch type)."""
    results = {}
    for case in CASES:
        prev = chain_from if chain_from is not None else results
        results[case[0]] = run_case(mods, case, scale, prev, device, dtype)
    return results


def load_golden(path):
    z = np.load(path)
    out = {}
    for case in CASES:
        k = 0
        outs = []
        while "%s__%d" % (case[0], k) in z.files:
            outs.append(torch.from_numpy(z["%s__%d" % (case[0], k)]))
            k += 1
        out[case[0]] = outs
    return out, z


def ulp_diff(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-2 ** 31) - ai, ai)
    bi = np.where(bi < 0, np.int64(-2 ** 31) - bi, bi)
    return np.abs(ai - bi)
