"""Parity of the fused render at the HEADLINE scale (BASELINE.json configs[1] / SURVEY 8d "S1": G = 200^3, F = 3, C = 12,
1920x1080 rays x 256 samples) -- VERDICT r1 "What's weak" #1.  131 072 rays (16 chunks of 8192 spread over the frame)
are compared with the CPU oracle, all rays, no margin carve-out.

* S1b (smooth fields, opaque surfaces: trained-like statistics, every ray ends on a surface): rgb, depth and
  alphainv_last within 1e-4 of the oracle on ALL rays -- in practice ~1e-7.
* S1 (white-noise grids, sigma = 24 density units per voxel): rgb AND depth within 1e-4 of the CPU reference on all rays, outright
  (round 5; measured 5.4e-5 / 8.2e-5 ... 9.8e-5); alphainv_last within max(1e-4, the reference's own distance to the fp64 truth).  A ray
  above the bound is admitted only under per-ray fp64 arbitration.  The tail near the bound is NOT produced by any arithmetic shortcut of
  the kernels (profiles/r02/parity_ab_s1.txt: IEEE divisions, libm sincos / pow and grid_sample's own corner sum change the worst rays
  by < 2e-6): it is the conditioning of the reference formula on this scene -- the reference executing on the MI355X (its own Python
  over its own compiled kernels, oracle/_ref) differs from its CPU evaluation by 1.4e-4 / 1.6e-4 / 2.8e-4, and against an fp64 ground
  truth the fused render is the closest of the three (test_s1_tail_against_fp64_ground_truth).
* truck_single.py's shape (F = 4 -> P = 9, S = 668) on the whole 1080p frame and viewbase_pe = 8 (bf16x3 kernel) at G = 100: 1e-4 on all
  sampled rays (measured 4e-7).
"""
import json
import os
import sys

import pytest
import torch

from oracle import model_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

G, H, W, N_CHUNKS, CHUNK, STEPSIZE = 200, 1080, 1920, 16, 8192, 1.31
KEYS = ("rgb_marched", "depth", "alphainv_last")


def reference_on_gpu(state, dev, ro, rd, vd, starts):
    """The reference ITSELF executing on this MI355X: its own FourierGridModel.forward
    (FourierGrid_model.py:554-672) over its own compiled kernels (oracle/_ref/<variant>/*.so, built from
    FourierGrid/cuda/*.cu by oracle/build_ref.py) and torch-ROCm grid_sample / Linear, in the 8192-ray chunks of its
    render loop (run_render.py:52-58).  Returns {variant: outputs on the sampled rays} or {} when oracle/_ref is absent."""
    from oracle import ref_model
    res = {}
    for variant in ("fma", "nofma"):
        if not ref_model.available("kernels:" + variant):
            continue
        model = ref_model.reference_model(state, dev, "kernels:" + variant)
        parts = [ref_model.render(model, ro[b:b + CHUNK], rd[b:b + CHUNK], vd[b:b + CHUNK], STEPSIZE, chunk=CHUNK) for b in starts]
        res[variant] = {k: torch.cat([p[k] for p in parts]).cpu() for k in KEYS}
        del model
        torch.cuda.empty_cache()
    return res


_CACHE = {}
_EXTRA = {}


def render_and_reference(make_state, two_libms, with_ref_gpu=False):
    """cached per (scene, options): the S1 headline test and the arbitration test share one evaluation"""
    key = (make_state.__name__, two_libms, with_ref_gpu)
    if key not in _CACHE:
        _CACHE[key] = _render_and_reference(make_state, two_libms, with_ref_gpu)
    return _CACHE[key]


def _render_and_reference(make_state, two_libms, with_ref_gpu=False):
    import bench
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_a_view
    dev = torch.device("cuda", 0)
    state = make_state(G, dev, seed=0)
    rend = FourierGridRenderer(state, dev)
    K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
    ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in get_rays_of_a_view(H, W, K, bench.camera(0, dev))]
    R = ro.shape[0]
    out = rend(ro, rd, vd, stepsize=STEPSIZE, render_depth=True)
    torch.cuda.synchronize()
    M = rend.survivors_of_last_chunk()
    starts = [int(i * (R - CHUNK) / (N_CHUNKS - 1)) // 64 * 64 for i in range(N_CHUNKS)]
    ref_gpu = reference_on_gpu(state, dev, ro, rd, vd, starts) if with_ref_gpu else {}
    cpu_state = {k: ([x.cpu() for x in v] if isinstance(v, list) else (v.cpu() if torch.is_tensor(v) else v)) for k, v in state.items()}
    del state, rend
    torch.cuda.empty_cache()
    idx = torch.cat([torch.arange(b, b + CHUNK) for b in starts])
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    refs = []
    for mode in (("torch", "rounded64") if two_libms else ("torch",)):
        model_oracle.PE_MATH = mode
        try:
            parts = [model_oracle.fouriergrid_render(cpu_state, ro[b:b + CHUNK].cpu(), rd[b:b + CHUNK].cpu(), vd[b:b + CHUNK].cpu(),
                                                     STEPSIZE, render_depth=True) for b in starts]
        finally:
            model_oracle.PE_MATH = "torch"
        refs.append({k: torch.cat([p[k] for p in parts]) for k in KEYS})
    got = {k: out[k].cpu()[idx] for k in KEYS}
    _EXTRA[make_state.__name__] = (cpu_state, (ro.cpu()[idx], rd.cpu()[idx], vd.cpu()[idx]))   # for the fp64 ground-truth study
    if with_ref_gpu:
        return got, refs, ref_gpu, M, R
    return got, refs, M, R


def per_ray_err(a, b):
    e = (a - b).abs()
    return e.amax(dim=1) if e.dim() == 2 else e


def test_s1b_surfaces_all_outputs_within_1e_4_on_all_rays():
    import bench
    got, (ref,), _, M, R = render_and_reference(bench.make_state_surfaces, two_libms=False, with_ref_gpu=True)
    assert float((ref["alphainv_last"] < 1e-3).float().mean()) > 0.3          # a surface scene: rays terminate early
    assert 0.01 < M / (R * 256.0) < 0.2
    # The reference stops a ray at the first sample whose transmittance drops below 1e-3 (render_utils_kernel.cu:452-455).
    # A ray whose T lands within an ulp of 1e-3 there stops on one side and goes on on the other: the outputs then differ
    # by what the rest of the ray still adds (< 1e-3) -- a discontinuity of the reference formula, not an error.  Such a
    # ray shows alphainv_last just below 1e-3 on the side that stopped and a much smaller value on the other.
    ga, ra = got["alphainv_last"], ref["alphainv_last"]
    at_stop = lambda x: (x > 0.999e-3) & (x < 1e-3)
    tie = (at_stop(ga) & (ra < 0.9e-3)) | (at_stop(ra) & (ga < 0.9e-3))
    assert int(tie.sum()) <= 3, int(tie.sum())
    for k in KEYS:
        err = per_ray_err(got[k], ref[k])
        print("S1b %-14s linf %.3e mean %.3e stop-threshold ties %d" % (k, float(err[~tie].max()), float(err.mean()), int(tie.sum())))
        assert float(err[~tie].max()) <= 1e-4, (k, float(err[~tie].max()))
        if bool(tie.any()):
            assert float(err[tie].max()) <= 1.1e-3, (k, float(err[tie].max()))


def test_s1_headline_scene_rgb_depth_parity():
    """S1 (white noise, sigma = 24 density units per voxel): rgb AND depth within the north-star 1e-4 of the CPU reference on all
    131 072 sampled rays, outright (round 4 measured 5.4e-5 / 8.2e-5 ... 9.8e-5); alphainv_last within max(1e-4, the CPU reference's
    own distance to the fp64 truth) with no more rays above 1e-4 than the reference itself has against the truth.  A ray above the
    bound is admitted only under per-ray fp64 arbitration (see `arbitrated` below).  Rounds 2-3 held depth / alphainv_last to "what
    the reference executing on this GPU differs from its CPU execution by" (1.6e-4 / 2.8e-4): VERDICT r4 weak #1 -- that bound would
    not notice a regression to 2e-4; the distance is still printed, it no longer bounds anything."""
    import bench
    from oracle import ref_model
    have_ref = ref_model.available("kernels:fma")
    got, (ref, ref2), ref_gpu, M, R = render_and_reference(bench.make_state, two_libms=True, with_ref_gpu=True)
    assert abs(M / (R * 256.0) - 0.049) < 0.003                              # the calibrated 4.9 % survivors
    stats = {}
    for k in KEYS:
        err = per_ray_err(got[k], ref[k])
        amb = per_ray_err(ref2[k], ref[k])                                   # two conforming libms, same formula
        rg = float(per_ray_err(ref_gpu["fma"][k], ref[k]).max()) if have_ref else None
        stats[k] = (float(err.max()), float(err.mean()), int((err > 1e-4).sum()), float(amb.max()), rg)
        print("S1  %-14s gpu-vs-oracle linf %.3e mean %.3e rays>1e-4 %d | oracle libm ambiguity linf %.3e | reference-on-GPU vs reference-on-CPU %s"
              % ((k,) + stats[k][:4] + (("%.3e" % rg) if rg is not None else "n/a",)))
    n = got["depth"].numel()
    for k in KEYS:
        assert stats[k][1] <= 5e-6, (k, stats[k][1])                         # mean error
    # The north-star bound, 1e-4 against the CPU reference, OUTRIGHT for rgb and depth (measured on the
    # round-4 boards: rgb 5.4e-5, depth 8.2e-5 ... 9.8e-5, 0 rays above).  The reference formula has three hard thresholds; a ray
    # on which one of two fp32 evaluations flips a threshold moves by up to the flipped sample's weight, so a ray just above the
    # bound is admitted only under fp64 arbitration, ray by ray: at most 3 such rays, none further than 1.25e-4, and on each the
    # fused value must be within 1e-4 of the fp64 truth or at least as close to it as the CPU reference is.  A drift of depth back
    # to 1.2e-4 on a handful of rays, or to 2e-4 on one, fails.
    from tools import parity_fp64
    cpu_state, (ro_s, rd_s, vd_s) = _EXTRA["make_state"]

    def arbitrated(k, cap_linf, max_rays):
        err = per_ray_err(got[k], ref[k])
        bad = torch.nonzero(err > 1e-4).flatten()
        if bad.numel() == 0:
            return 0
        assert bad.numel() <= max_rays and float(err.max()) <= cap_linf, (k, int(bad.numel()), float(err.max()))
        truth = parity_fp64.render_fp64(cpu_state, ro_s[bad], rd_s[bad], vd_s[bad], STEPSIZE)
        d_f, d_r = per_ray_err(got[k][bad].double(), truth[k]), per_ray_err(ref[k][bad].double(), truth[k])
        print("S1  %-14s %d ray(s) above 1e-4 vs the CPU reference, arbitrated in fp64: |fused - truth| %s, |reference - truth| %s"
              % (k, bad.numel(), ["%.2e" % x for x in d_f.tolist()], ["%.2e" % x for x in d_r.tolist()]))
        assert bool(((d_f <= 1e-4) | (d_f <= d_r + 2e-5)).all()), (k, d_f.tolist(), d_r.tolist())
        return int(bad.numel())

    assert stats["rgb_marched"][0] <= 1e-4, stats["rgb_marched"]
    arbitrated("depth", 1.25e-4, 3)
    # alphainv_last (the product of ~250 factors, the scene's most sensitive output): <= max(1e-4, what the CPU reference ITSELF is
    # away from the fp64 truth on these rays), and no more rays above 1e-4 than the reference has against the truth
    study = _fp64_study()
    ref_truth = study["distance_to_fp64"]["ref_cpu"]["alphainv_last"]
    linf, _, n_above = stats["alphainv_last"][:3]
    print("S1  alphainv_last    linf %.3e (%d rays > 1e-4) against the bound max(1e-4, CPU reference vs fp64 = %.3e, %d rays > 1e-4)"
          % (linf, n_above, ref_truth["linf"], ref_truth["rays_above_bound"]))
    assert linf <= max(1e-4, ref_truth["linf"]), (linf, ref_truth)
    assert n_above <= ref_truth["rays_above_bound"], (n_above, ref_truth)
    arbitrated("alphainv_last", max(1e-4, ref_truth["linf"]), max(1, ref_truth["rays_above_bound"]))
    # what the reference's own execution on this GPU differs from its CPU execution by stays in the printout above (stats[k][4]); it
    # is no longer a bound (it is 1.4e-4 / 1.6e-4 / 2.8e-4: held to it, the test would not notice a regression to 2e-4)


_STUDY = {}


def _fp64_study():
    """tools/parity_fp64.ground_truth_study of the S1 evaluation, once per session (the headline test and the tail test share it)"""
    if "s1" not in _STUDY:
        import bench
        from tools import parity_fp64
        got, (ref, _), ref_gpu, M, R = render_and_reference(bench.make_state, two_libms=True, with_ref_gpu=True)
        cpu_state, rays = _EXTRA["make_state"]
        evals = {"fused": got, "ref_cpu": ref}
        if "fma" in ref_gpu:
            evals["ref_gpu"] = ref_gpu["fma"]
        torch.set_num_threads(min(8, os.cpu_count() or 1))
        _STUDY["s1"] = (parity_fp64.ground_truth_study(cpu_state, rays, evals, STEPSIZE, n_random=1024, seed=0), evals)
    return _STUDY["s1"][0]


def test_s1_tail_against_fp64_ground_truth():
    """the S1 tail judged against a GROUND TRUTH instead of a carve-out.  Every sampled ray where the fused
    render is further than 1e-4 from the CPU reference or from the reference executing on this GPU, plus 1024 random rays, is
    re-evaluated in fp64 (tools/parity_fp64.py: the same formula, thresholds and sample table, every tensor and libm call in
    double).  On those rays
      * the north-star bound holds against the truth wherever the fp32 reference itself is sound: on every ray where a
        reference is within 1e-4 of fp64 in all three outputs, so is the fused render;
      * the fused render is never further from the truth than a reference by more than the 2e-5 slack on more than a handful
        of rays (a threshold flip moves ONE of two fp32 evaluations: either side can be the lucky one, and the flip moves
        the ray by up to the flipped sample's weight), and on average it is no further from the truth than the references are.
    The numbers go to gpurun_out/s1_fp64_ground_truth.json (committed as profiles/r04/s1_fp64_ground_truth.json)."""
    import bench
    from oracle import ref_model
    from tools import parity_fp64
    res = _fp64_study()
    evals = _STUDY["s1"][1]
    _dump("s1_fp64_ground_truth.json", res)
    for name, d in res["distance_to_fp64"].items():
        print("S1 fp64  %-8s " % name + "  ".join("%s linf %.3e mean %.2e (%d > 1e-4)" % (k[:5], d[k]["linf"], d[k]["mean_abs"], d[k]["rays_above_bound"]) for k in KEYS))
    fused = res["distance_to_fp64"]["fused"]
    for name in evals:
        if name == "fused":
            continue
        ok, viol, ref = res["bound_where_ref_ok"][name], res["violations"][name], res["distance_to_fp64"][name]
        print("S1 fp64  vs %-8s fused linf where that reference is within 1e-4 of fp64 (%d rays): %s | rays where fused / the reference is the one "
              "further from fp64 by > 2e-5: %s" % (name, ok["rays_where_ref_within_bound_of_fp64"], {k: ok[k] for k in KEYS},
                                                   {k: (viol[k]["rays"], viol[k]["rays_where_the_reference_is_further_than_fused"]) for k in KEYS}))
        for k in KEYS:
            # (1) the fused render's error against the truth is no larger than the reference's own: worst ray, mean, tail size
            assert fused[k]["linf"] <= 1.1 * ref[k]["linf"] + 1e-5, (name, k, fused[k], ref[k])
            assert fused[k]["mean_abs"] <= 1.1 * ref[k]["mean_abs"] + 1e-7, (name, k, fused[k], ref[k])
            assert fused[k]["rays_above_bound"] <= ref[k]["rays_above_bound"] + 3, (name, k, fused[k], ref[k])
            # (2) a threshold flip moves ONE of two fp32 evaluations: neither side is systematically the unlucky one
            assert viol[k]["rays"] <= viol[k]["rays_where_the_reference_is_further_than_fused"] + 6, (name, k, viol[k])
    # (3) the north-star bound against the truth: on every ray where the CPU reference (the pinned restatement of the reference's
    # own arithmetic) is within 1e-4 of fp64 the fused render is too (10 % allowance: the rays sit on flipped thresholds, where the
    # truth itself is one side of a discontinuity), and on the rays where every fp32 reference is sound (within 5e-5) it is within
    # 1e-4 with margin
    okc = res["bound_where_ref_ok"]["ref_cpu"]
    snd = res["fused_on_rays_where_all_references_are_within_half_bound"]
    print("S1 fp64  fused on the %d rays where every reference is within 5e-5 of fp64: %s" % (snd["rays"], {k: snd[k] for k in KEYS}))
    for k in KEYS:
        assert okc[k] is not None and okc[k] <= 1.1e-4, (k, okc[k])
        assert snd["rays"] >= 800 and snd[k] <= 1e-4, (k, snd)


def pairwise_table(named):
    """L-inf / rays above 1e-4 / mean for every pair of evaluations, per output"""
    names = list(named)
    table = {}
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            row = {}
            for k in KEYS:
                err = per_ray_err(named[a][k], named[b][k])
                row[k] = {"linf": float(err.max()), "rays_above_1e-4": int((err > 1e-4).sum()), "mean_abs": float(err.mean())}
            table["%s <-> %s" % (a, b)] = row
    return table


def _dump(name, obj):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("scene", ["s1", "s1b"])
def test_arbitration_against_the_reference_executing_on_this_gpu(scene):
    """VERDICT r2 "next round" item 1.  Three evaluations of the same 131 072 rays of the headline frame:
      fused      -- this library (k_march + k_shade_mlp)
      ref-gpu    -- the reference's own Python + its own compiled kernels + torch-ROCm grid_sample on this MI355X
      cpu-oracle -- oracle/model_oracle.py (torch CPU restatement, pinned bit for bit on the reference's Python)
    The pairwise L-inf table is printed and written to gpurun_out/s1_arbitration_<scene>.json.  The fused render has to
    be as close to the reference-on-GPU as the reference-on-GPU is to the reference-on-CPU (two executions of the SAME
    code on different hardware), and within the north-star 1e-4 wherever those two agree to 1e-4 themselves."""
    import bench
    from oracle import ref_model
    if not ref_model.available("kernels:fma"):
        pytest.skip("oracle/_ref (compiled reference kernels + reference_py.tar) not staged: python oracle/build_ref.py")
    if scene == "s1":
        got, (ref, _), ref_gpu, M, R = render_and_reference(bench.make_state, two_libms=True, with_ref_gpu=True)
    else:
        got, (ref,), ref_gpu, M, R = render_and_reference(bench.make_state_surfaces, two_libms=False, with_ref_gpu=True)
    named = {"fused": got, "cpu-oracle": ref}
    for variant, o in ref_gpu.items():
        named["ref-gpu(%s)" % variant] = o
    table = pairwise_table(named)
    for pair, row in table.items():
        print("%-4s %-34s " % (scene.upper(), pair) + "  ".join("%s %.3e (%d > 1e-4)" % (k[:5], row[k]["linf"], row[k]["rays_above_1e-4"]) for k in KEYS))
    _dump("s1_arbitration_%s.json" % scene, {"scene": scene, "rays": int(got["depth"].numel()), "survivors_frame": M, "pairs": table})
    # stop-threshold ties (see test_s1b_...): a ray whose T lands within an ulp of 1e-3 stops on one side only
    at_stop = lambda x: (x > 0.999e-3) & (x < 1e-3)
    for variant in ref_gpu:
        rg = ref_gpu[variant]
        tie = (at_stop(got["alphainv_last"]) & (rg["alphainv_last"] < 0.9e-3)) | (at_stop(rg["alphainv_last"]) & (got["alphainv_last"] < 0.9e-3))
        assert int(tie.sum()) <= 3
        for k in KEYS:
            fused_vs_ref = float(per_ray_err(got[k], rg[k])[~tie].max())
            ref_vs_cpu = float(per_ray_err(rg[k], ref[k]).max())
            # the reference against itself (GPU vs CPU execution) bounds what any third implementation can be held to
            assert fused_vs_ref <= max(1e-4, ref_vs_cpu + 2e-5), (scene, variant, k, fused_vs_ref, ref_vs_cpu)


def test_s5_block_shape_g300_l2_c3_pe2_parity():
    """BASELINE configs[4] at its real block shape (configs/waymo/waymo_no_block.py:129-149): G = 300^3, rgbnet_dim = 3,
    viewbase_pe = 2, contracted_norm = 'l2', stepsize 0.5 (S = 1002) -- 16 384 rays spread over the frame against the
    CPU oracle, all three outputs within 1e-4 on all rays; then the same renderer through dist.composite_blocks (the
    one-block-per-GPU compositing path, single block here) must return the block's own image."""
    import bench
    from unboundednerfpytorch_amd.dist import composite_blocks
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_a_view
    dev = torch.device("cuda", 0)
    state = bench.make_state_surfaces(300, dev, seed=0, C=3, pe=2, norm="l2")
    rend = FourierGridRenderer(state, dev)
    K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
    c2w = bench.camera(0, dev)
    ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in get_rays_of_a_view(H, W, K, c2w)]
    R = ro.shape[0]
    starts = [int(i * (R - 4096) / 3) // 64 * 64 for i in range(4)]
    idx = torch.cat([torch.arange(b, b + 4096) for b in starts]).to(dev)
    o, d, v = ro[idx].contiguous(), rd[idx].contiguous(), vd[idx].contiguous()
    out = rend(o, d, v, stepsize=0.5, render_depth=True)
    assert out["n_max"] == 1002
    cpu_state = {k: ([x.cpu() for x in v_] if isinstance(v_, list) else (v_.cpu() if torch.is_tensor(v_) else v_)) for k, v_ in state.items()}
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ref = model_oracle.fouriergrid_render(cpu_state, o.cpu(), d.cpu(), v.cpu(), 0.5, render_depth=True)
    assert float((ref["alphainv_last"] < 1e-3).float().mean()) > 0.3
    for k in KEYS:
        err = per_ray_err(out[k].cpu(), ref[k])
        print("S5  %-14s linf %.3e mean %.3e" % (k, float(err.max()), float(err.mean())))
        assert float(err.max()) <= 1e-4, (k, float(err.max()))
    comp = composite_blocks(rend.forward, o, d, v, c2w[:, 3].tolist(), [0.5, 0.0, 0.0], stepsize=0.5)
    for k in KEYS:                      # (w * x) / w of the merging rule: equal up to that rounding
        assert float((comp[k] - out[k]).abs().max()) <= 1e-6, k
    assert comp["block_weight"] > 0


def _frame_rays(dev):
    import bench
    from unboundednerfpytorch_amd.fourier_render import get_rays_of_a_view
    K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
    return [x.reshape(-1, 3).contiguous() for x in get_rays_of_a_view(H, W, K, bench.camera(0, dev))]


def _cpu_state(state):
    return {k: ([x.cpu() for x in v_] if isinstance(v_, list) else (v_.cpu() if torch.is_tensor(v_) else v_)) for k, v_ in state.items()}


def test_truck_shape_f4_p9_g200_s668_frame_parity():
    """BASELINE configs[2]'s RENDER half at its real shape (VERDICT r4 "missing" #1; configs/tankstemple_unbounded/truck_single.py:92-110:
    fourier_freq_num = 4 -> P = 9, G = 200^3, rgbnet_dim 12, stepsize 0.5 -> S = 668; frame loop run_render.py:54-66).  F >= 4 selects the
    8-wave producer / consumer shade kernel (k_shade_pc<4,4,4,4,NBL,0>), which no other test runs above G = 33.  The WHOLE 1920x1080
    frame is rendered in 8 x 8 pixel blocks (what render_view and the bench do); 32 768 rays spread over it are held to 1e-4 against the
    CPU oracle in all three outputs, on all sampled rays, and the same rays rendered on their own must give the same bits (per-ray
    results do not depend on the chunking, the tile a ray sits in, or the work-list size)."""
    import bench
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, pixel_tile_order, untile
    dev = torch.device("cuda", 0)
    state = bench.make_state_surfaces(G, dev, seed=0, F=4)
    rend = FourierGridRenderer(state, dev)
    assert rend.F == 4 and rend.mlp_mode == 2                      # fp16x2 rgbnet -> the producer / consumer kernel
    ro, rd, vd = _frame_rays(dev)
    R = ro.shape[0]
    order = pixel_tile_order(H, W, dev)
    out_t = rend(ro[order].contiguous(), rd[order].contiguous(), vd[order].contiguous(), stepsize=0.5, render_depth=True, ray_order="coherent")
    assert out_t["n_max"] == 668
    M = rend.survivors_of_last_chunk()
    out = {k: untile(out_t[k], H, W) for k in KEYS}
    torch.cuda.synchronize()
    n_ck, ck = 4, 8192
    starts = [int(i * (R - ck) / (n_ck - 1)) // 64 * 64 for i in range(n_ck)]
    idx = torch.cat([torch.arange(b, b + ck) for b in starts]).to(dev)
    o, d, v = ro[idx].contiguous(), rd[idx].contiguous(), vd[idx].contiguous()
    sub = rend(o, d, v, stepsize=0.5, render_depth=True, ray_order="coherent")
    for k in KEYS:
        assert torch.equal(sub[k], out[k][idx]), "per-ray results depend on the ray list (%s)" % k
    cpu_state = _cpu_state(state)
    del state, rend
    torch.cuda.empty_cache()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ref = model_oracle.fouriergrid_render(cpu_state, o.cpu(), d.cpu(), v.cpu(), 0.5, render_depth=True)
    term = float((ref["alphainv_last"] < 1e-3).float().mean())
    print("truck F=4 P=9 G=200 S=668: survivors %.2f M (%.2f %% of R x S), %.0f %% of the sampled rays end on a surface" % (M / 1e6, 100.0 * M / (R * 668.0), 100 * term))
    assert term > 0.3 and 0.005 < M / (R * 668.0) < 0.2
    ga, ra = sub["alphainv_last"].cpu(), ref["alphainv_last"]
    at_stop = lambda x: (x > 0.999e-3) & (x < 1e-3)              # a ray whose T lands within an ulp of the 1e-3 stop (see the S1b test)
    tie = (at_stop(ga) & (ra < 0.9e-3)) | (at_stop(ra) & (ga < 0.9e-3))
    assert int(tie.sum()) <= 3, int(tie.sum())
    res = {}
    for k in KEYS:
        err = per_ray_err(sub[k].cpu(), ref[k])
        res[k] = {"linf": float(err[~tie].max()), "mean_abs": float(err.mean()), "rays_above_1e-4": int((err[~tie] > 1e-4).sum())}
        print("truck %-14s linf %.3e mean %.3e stop-threshold ties %d" % (k, res[k]["linf"], res[k]["mean_abs"], int(tie.sum())))
        assert res[k]["linf"] <= 1e-4, (k, res[k])
    _dump("truck_f4_frame_parity.json", {"rays": int(idx.numel()), "survivors_frame": M, "outputs": res})


def test_frame_loop_two_views_in_flight_equals_one_stream_at_1080p():
    """Row f1 at frame scale (run_render.py:54-66): run_render.render_viewpoints over four 1080p views of the trained-like scene with
    consecutive views alternating between two streams / two work lists (the default: the march of view k + 1 runs beside the shade of
    view k) returns, bit for bit, what the one-stream loop returns and what a stand-alone render_view of each pose returns."""
    import numpy as np
    import bench
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer
    from unboundednerfpytorch_amd.run_render import render_viewpoints
    dev = torch.device("cuda", 0)
    rend = FourierGridRenderer(bench.make_state_surfaces(G, dev, seed=0), dev)
    K = np.array([[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]])
    poses = [bench.camera(i, dev).cpu().numpy() for i in range(4)]
    kw = {"stepsize": 0.5, "inverse_y": False}
    two = render_viewpoints(rend, poses, [(H, W)] * 4, [K] * 4, kw)
    one = render_viewpoints(rend, poses, [(H, W)] * 4, [K] * 4, kw, frames_in_flight=1)
    for a, b in zip(two, one):
        assert a.shape[:3] == (4, H, W) and np.array_equal(a, b)
    assert not np.array_equal(two[0][0], two[0][1]) and np.isfinite(two[0]).all()
    r, d, b = rend.render_view(H, W, K, poses[3], 0.5)
    assert np.array_equal(two[0][3], r.cpu().numpy()) and np.array_equal(two[1][3][..., 0], d.cpu().numpy())
    assert np.array_equal(two[2][3][..., 0], b.cpu().numpy())
    assert float((two[2] < 1e-3).mean()) > 0.2                # (the scene has surfaces: rays end on them)
    del rend
    torch.cuda.empty_cache()


def test_viewbase_pe8_bf16x3_g100_parity():
    """viewbase_pe = 8 (configs/waymo/waymo_base.py, configs/mega/*.py): the 51-wide view embedding does not fit the fp16x2 kernels'
    LDS budget, ugrid_pack_mlp reports bf16x3 and k_shade_mlp<3,12,8,8,1> renders it -- checked so far on small goldens only.
    16 384 rays of the 1080p view at G = 100, stepsize 0.5, all three outputs within 1e-4 of the CPU oracle on all rays."""
    import bench
    from unboundednerfpytorch_amd import _lib
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer
    dev = torch.device("cuda", 0)
    state = bench.make_state_surfaces(100, dev, seed=0, C=12, pe=8)
    rend = FourierGridRenderer(state, dev)
    assert rend.pe == 8 and rend.mlp_mode == _lib.MLP_BF16X3
    ro, rd, vd = _frame_rays(dev)
    R = ro.shape[0]
    starts = [int(i * (R - 4096) / 3) // 64 * 64 for i in range(4)]
    idx = torch.cat([torch.arange(b, b + 4096) for b in starts]).to(dev)
    o, d, v = ro[idx].contiguous(), rd[idx].contiguous(), vd[idx].contiguous()
    out = rend(o, d, v, stepsize=0.5, render_depth=True, ray_order="coherent")
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ref = model_oracle.fouriergrid_render(_cpu_state(state), o.cpu(), d.cpu(), v.cpu(), 0.5, render_depth=True)
    assert float((ref["alphainv_last"] < 1e-3).float().mean()) > 0.3
    for k in KEYS:
        err = per_ray_err(out[k].cpu(), ref[k])
        print("pe8  %-14s linf %.3e mean %.3e" % (k, float(err.max()), float(err.mean())))
        assert float(err.max()) <= 1e-4, (k, float(err.max()))
