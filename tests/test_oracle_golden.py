"""Pins the oracle (oracle/model_oracle.py + oracle/ref_ops.c) against the golden vectors produced
by the reference's own Python code (tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import model_oracle, ref_ops

FG_CASES = [
    # name, seed, G, F, C, pe, norm, stepsize, R, thres, dens_mean, dens_std   (== gen_golden.FG_CASES)
    ("fg_inf_f3_c12", 11, 16, 3, 12, 4, "inf", 0.5, 96, 1e-4, -3.0, 8.0),
    ("fg_l2_f2_c3", 12, 12, 2, 3, 2, "l2", 0.7, 64, 1e-4, -2.0, 6.0),
    ("fg_inf_f4_c12_dense", 13, 10, 4, 12, 4, "inf", 0.5, 48, 1e-4, 8.0, 12.0),
    ("fg_norgbnet", 14, 12, 3, 0, 4, "inf", 0.5, 64, 1e-4, 5.0, 12.0),
    ("fg_inf_f3_c12_medium", 15, 14, 3, 12, 4, "inf", 1.31, 80, 1e-4, 4.0, 12.0),
]


def make_state(seed, G, F, C, pe, norm, thres, dm, ds, width=128):
    """Plain-tensor model state equivalent to FourierGridModel(xyz_min=-1, xyz_max=1, num_voxels=G^3,
    alpha_init=1e-4, bg_len=0.2) with synthetic parameters."""
    p = synth.fouriergrid_params(seed, G, F, C, width=width, viewbase_pe=pe, dens_mean=dm, dens_std=ds)
    names = ['rgbnet.0'] + ['rgbnet.%d.0' % i for i in range(2, 3)] + ['rgbnet.3']
    ws = [torch.from_numpy(p[n + '.weight']) for n in names] if C > 0 else []
    bs = [torch.from_numpy(p[n + '.bias']) for n in names] if C > 0 else []
    return {
        'density_grid': torch.from_numpy(p['density.grid']), 'k0_grid': torch.from_numpy(p['k0.grid']),
        'rgbnet_weights': ws, 'rgbnet_biases': bs,
        'scene_center': torch.zeros(3), 'scene_radius': torch.ones(3),
        'xyz_min': torch.full((3,), -1.2), 'xyz_max': torch.full((3,), 1.2),
        'bg_len': 0.2, 'fourier_freq_num': F, 'viewbase_pe': pe,
        'act_shift': float(model_oracle.act_shift_from_alpha_init(1e-4)), 'voxel_size_ratio': 1.0,
        'fast_color_thres': thres, 'contracted_norm': norm, 'world_len': G,
    }


@pytest.mark.parametrize("case", FG_CASES, ids=[c[0] for c in FG_CASES])
def test_fouriergrid_render_matches_reference(case, golden_dir):
    name, seed, G, F, C, pe, norm, stepsize, R, thres, dm, ds = case
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    torch.set_num_threads(1)
    state = make_state(seed, G, F, C, pe, norm, thres, dm, ds)
    o, d, v = [torch.from_numpy(a) for a in synth.rays(seed, R)]
    out = model_oracle.fouriergrid_render(state, o, d, v, stepsize, render_depth=True)
    assert out['n_max'] == int(gold['n_max'])
    # index outputs: bit-exact
    assert np.array_equal(out['ray_id'].numpy(), gold['ray_id'])
    assert np.array_equal(out['step_id'].numpy(), gold['step_id'])
    # float outputs: the restatement issues the same torch ops in the same order.  In the container that
    # generated the fixtures this is bit-exact; another host CPU (other SIMD width / thread count on the GPU
    # box) changes torch's sin/exp/sgemm kernels by <= 1 ulp, hence a 2e-6 relative tolerance.
    for k in ("alphainv_last", "weights", "rgb_marched", "raw_density", "raw_alpha", "raw_rgb", "t", "s", "depth"):
        np.testing.assert_allclose(out[k].numpy(), gold[k], rtol=2e-6, atol=2e-9, err_msg=k)


def test_grid_query_matches_reference(golden_dir):
    gold = np.load(os.path.join(golden_dir, "grid_query.npz"))
    n = 257
    pts = torch.from_numpy(synth.uniform(31, n * 3, -1.5, 1.5).reshape(n, 3))
    pts[:8] = torch.tensor([[-1.2, -1.2, -1.2], [1.2, 1.2, 1.2], [0, 0, 0], [1.2, -1.2, 0.3],
                            [1.3, 0, 0], [0, -1.25, 0], [0.1, 0.2, 1.2000001], [-1.2, 1.2, -1.2]])
    lo, hi = torch.full((3,), -1.2), torch.full((3,), 1.2)
    for C, F in ((1, 3), (12, 3), (3, 2)):
        G = (9, 7, 5)
        g = torch.from_numpy(synth.normal(40 + C, (1 + 2 * F) * C * G[0] * G[1] * G[2]).reshape(1 + 2 * F, C, *G))
        got = model_oracle.fourier_grid_query(g, pts, lo, hi, F)
        np.testing.assert_allclose(got.numpy(), gold["fourier_c%d_f%d" % (C, F)], rtol=2e-6, atol=1e-6)
    lo, hi = torch.tensor([-1.0, -0.5, -2.0]), torch.tensor([1.0, 1.5, 1.0])
    for C in (1, 4):
        G = (6, 8, 11)
        g = torch.from_numpy(synth.normal(50 + C, C * G[0] * G[1] * G[2]).reshape(1, C, *G))
        got = model_oracle.fourier_grid_query(g, pts, lo, hi, 0)
        np.testing.assert_array_equal(got.numpy(), gold["dense_c%d" % C])


def test_oracle_ops_invariants():
    """Known-answer / property checks of the C restatement (the reference holds no vectors for it)."""
    # alpha2weight: sum(w) + alphainv_last == 1 for rays that never hit the early stop
    n, R = 1000, 40
    alpha = torch.from_numpy(synth.uniform(5, n, 0.0, 0.05))
    ray_id = torch.from_numpy(np.sort((synth.uniform(6, n) * R).astype(np.int64)))
    w, T, last, i_s, i_e = ref_ops.alpha2weight(alpha, ray_id, R)
    tot = torch.zeros(R).index_add_(0, ray_id, w) + last
    assert torch.all(last > 1e-3)
    np.testing.assert_allclose(tot.numpy(), 1.0, atol=2e-6)
    # early stop: the crossing sample keeps its weight, later ones get w=0, T=1 (render_utils_kernel.cu:597)
    alpha = torch.tensor([0.5, 0.9, 0.99, 0.5, 0.5, 0.3])
    ray_id = torch.tensor([0, 0, 0, 0, 0, 2])
    w, T, last, i_s, i_e = ref_ops.alpha2weight(alpha, ray_id, 3)
    assert w[2] > 0 and w[3] == 0 and w[4] == 0 and T[3] == 1 and T[4] == 1
    assert i_e[0] == 3 and last[1] == 1 and i_s[1] == 0 and i_e[1] == 0
    assert abs(float(last[0]) - 0.5 * 0.1 * 0.01) < 1e-9 and float(last[2]) == pytest.approx(0.7)
    # maskcache: C round() is half away from zero (torch.round is half-to-even)
    world = torch.zeros(4, 4, 4, dtype=torch.bool)
    world[1, 3, 0] = True
    xyz = torch.tensor([[0.5, 2.5, -0.4], [1.49, 3.4, 0.49], [4.0, 0, 0], [-0.6, 0, 0]])
    out = ref_ops.maskcache_lookup(world, xyz, torch.ones(3), torch.zeros(3))
    assert out.tolist() == [True, True, False, False]
    # raw2alpha: exp overflow -> alpha exactly 1, backward clamps exp at 1e10
    e, a = ref_ops.raw2alpha(torch.tensor([200.0, -200.0]), 0.0, 0.5)
    assert torch.isinf(e[0]) and a[0] == 1 and a[1] == 0
    # masked adam leaves zero-grad entries bit-identical
    p = torch.from_numpy(synth.normal(7, 64)); p0 = p.clone()
    g = torch.from_numpy(synth.normal(8, 64)); g[::2] = 0
    m = torch.zeros(64); v = torch.zeros(64)
    ref_ops.masked_adam_upd(p, g, m, v, 1, 0.9, 0.99, 0.1, 1e-8)
    assert torch.equal(p[::2], p0[::2]) and torch.all(p[1::2] != p0[1::2]) and torch.all(m[::2] == 0)
    # first Adam step moves every touched entry by ~lr
    np.testing.assert_allclose((p0 - p)[1::2].abs().numpy(), 0.1, rtol=1e-4)
    # TV quirk: wx is ignored, the x-axis term uses wz (total_variation_kernel.cu:31-32)
    prm = torch.from_numpy(synth.normal(9, 2 * 3 * 4 * 5).reshape(1, 2, 3, 4, 5))
    g1 = torch.zeros_like(prm); g2 = torch.zeros_like(prm)
    ref_ops.total_variation_add_grad(prm, g1, 1.0, 2.0, 3.0, True)
    ref_ops.total_variation_add_grad(prm, g2, 77.0, 2.0, 3.0, True)
    assert torch.equal(g1, g2) and g1.abs().sum() > 0
    # cumdist: running sum resets only when it exceeds the threshold
    mask = ref_ops.cumdist_thres(torch.tensor([[0.4, 0.4, 0.4, 0.4, 0.4]]), 1.0)
    assert mask.tolist() == [[False, False, True, False, False]]


def test_distortion_loss_pinned_on_the_reference_class(golden_dir):
    """DistortionLoss (FourierGrid_model.py:684-708): tests/golden/distortion.npz was produced by the reference's own
    class (its segment_cumsum call served by the oracle op, which the reference extension never shipped).  Here the
    oracle op is checked against an independent fp64 evaluation and the golden loss / gradient against the
    definition  (1/R) [ sum_{i != j in a ray} w_i w_j |s_i - s_j| + (1/(3 n_max)) sum w_i^2 ]  differentiated by
    torch autograd in fp64."""
    gold = np.load(os.path.join(golden_dir, "distortion.npz"))
    w, s, ray_id, n_max = synth.distortion_inputs()
    pre = ref_ops.segment_cumsum(torch.from_numpy(w), torch.from_numpy(s), torch.from_numpy(ray_id))
    for got, key in zip(pre, ("w_prefix", "w_total", "ws_prefix", "ws_total")):
        assert np.array_equal(got.numpy(), gold[key]), key
    R = int(ray_id.max()) + 1
    w64, s64 = w.astype(np.float64), s.astype(np.float64)
    for r in range(R):
        m = ray_id == r
        cw = np.concatenate([[0.0], np.cumsum(w64[m])])
        cws = np.concatenate([[0.0], np.cumsum(w64[m] * s64[m])])
        np.testing.assert_allclose(gold["w_prefix"][m], cw[:-1], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(gold["ws_prefix"][m], cws[:-1], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(gold["w_total"][r], cw[-1], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(gold["ws_total"][r], cws[-1], rtol=2e-6, atol=1e-7)
    wt = torch.from_numpy(w64).requires_grad_(True)
    st = torch.from_numpy(s64)
    total = 0.0
    for r in range(R):
        m = torch.from_numpy(ray_id == r)
        if int(m.sum()) == 0:
            continue
        wr, sr = wt[m], st[m]
        total = total + (wr[:, None] * wr[None, :] * (sr[:, None] - sr[None, :]).abs()).sum() + (wr ** 2).sum() / (3 * n_max)
    loss = total / R
    loss.backward()
    np.testing.assert_allclose(float(gold["loss"]), float(loss), rtol=5e-6)
    # reference quirk, kept: the forward divides by n_rays but DistortionLoss.backward (FourierGrid_model.py:699-708)
    # does not, so the class returns n_rays x the true gradient of the loss it reports
    np.testing.assert_allclose(gold["grad"], R * wt.grad.numpy(), rtol=2e-5, atol=2e-6)


def test_train_forward_backward_pinned_on_the_reference_model(golden_dir):
    """tests/golden/train_step.npz holds loss, outputs and the gradient of EVERY parameter from one training forward
    + backward of the reference's own FourierGridModel (forward(global_step=1, is_train=True), run_train.py loss).
    The oracle's restatement of that training forward -- the CPU side of tests/test_gpu_train_step.py -- must
    reproduce them (same torch ops, same oracle extension ops: bit-exact in the generating container, 2e-6
    relative elsewhere)."""
    c = synth.TRAIN_CASE
    gold = np.load(os.path.join(golden_dir, "train_step.npz"))
    torch.set_num_threads(1)
    cfg = make_state(c["seed"], c["G"], c["F"], c["C"], c["pe"], c["norm"], c["thres"], c["dm"], c["ds"])
    params = {'density_grid': cfg['density_grid'].clone(), 'k0_grid': cfg['k0_grid'].clone(),
              'w0': cfg['rgbnet_weights'][0].clone(), 'b0': cfg['rgbnet_biases'][0].clone(),
              'w1': cfg['rgbnet_weights'][1].clone(), 'b1': cfg['rgbnet_biases'][1].clone(),
              'w2': cfg['rgbnet_weights'][2].clone(), 'b2': cfg['rgbnet_biases'][2].clone()}
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    Raw2Alpha, Alphas2Weights = model_oracle.make_autograd_ops(ref_ops)
    o, d, v = [torch.from_numpy(a) for a in synth.rays(c["seed"], c["R"])]
    target = torch.from_numpy(synth.uniform(c["seed"] + 5, c["R"] * 3).reshape(c["R"], 3))
    out = model_oracle.fouriergrid_train_forward(params, cfg, o, d, v, c["stepsize"], Raw2Alpha, Alphas2Weights)
    loss = torch.nn.functional.mse_loss(out['rgb_marched'], target)
    pout = out['alphainv_last'].clamp(1e-6, 1 - 1e-6)
    loss = loss + 0.01 * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout))).mean()
    loss.backward()
    assert out['n_kept'] == int(gold['n_kept'])
    np.testing.assert_allclose(float(loss), float(gold['loss']), rtol=2e-6)
    np.testing.assert_allclose(out['rgb_marched'].detach().numpy(), gold['rgb_marched'], rtol=2e-6, atol=1e-7)
    names = {'density_grid': 'density.grid', 'k0_grid': 'k0.grid', 'w0': 'rgbnet.0.weight', 'b0': 'rgbnet.0.bias',
             'w1': 'rgbnet.2.0.weight', 'b1': 'rgbnet.2.0.bias', 'w2': 'rgbnet.3.weight', 'b2': 'rgbnet.3.bias'}
    for k, ref_name in names.items():
        g = gold['grad.' + ref_name]
        got = params[k].grad.numpy()
        assert got.shape == g.shape, k
        scale = np.abs(g).max()
        assert np.abs(got - g).max() <= 2e-6 * scale + 1e-12, (k, float(np.abs(got - g).max()), float(scale))
        assert np.array_equal(got == 0, g == 0), k      # the touched-voxel mask MaskedAdam keys on


def test_c_oracle_pinned_on_reference_kernels(golden_dir):
    """oracle/ref_ops.c against outputs of the REFERENCE'S OWN native kernels (FourierGrid/cuda/*.cu compiled by
    oracle/build_ref.py, run on an MI355X by tests/golden/gen_native_golden.py): all 18 exported functions on the
    seeded cases of tests/native_cases.py.  Bit-exact for everything that is integer / IEEE add-mul-div-sqrt work;
    the raw2alpha family goes through expf / powf (glibc here, device libm there): exp <= 2 ulp, alpha <= 1.2e-7
    absolute (the subtraction from 1 amplifies pow's ulp for small alpha), gradients <= 4 ulp."""
    import native_cases as nc
    gold, z = nc.load_golden(os.path.join(golden_dir, "native_ops.npz"))
    assert sum(nc.EXPORTED.values()) == 18 and len({(c[1], c[2]) for c in nc.CASES}) == 18   # every m.def covered
    orc = {nc.RU: ref_ops.render_utils_cuda, nc.TV: ref_ops.total_variation_cuda, nc.UB: ref_ops.ub360_utils_cuda,
           nc.AD: ref_ops.adam_upd_cuda}
    got = nc.run_all(orc, scale=1, chain_from=gold)
    for name in gold:
        assert len(got[name]) == len(gold[name]) > 0, name
        for k, (a, b) in enumerate(zip(gold[name], got[name])):
            key = "%s[%d]" % (name, k)
            assert a.shape == b.shape and a.dtype == b.dtype, key
            if name not in nc.TRANSCENDENTAL:
                assert np.array_equal(a.numpy(), b.numpy(), equal_nan=True), key
                continue
            an, bn = a.numpy(), b.numpy()
            assert np.array_equal(np.isfinite(an), np.isfinite(bn)), key
            fin = np.isfinite(an)
            assert np.array_equal(an[~fin], bn[~fin]), key                     # +inf where the exponent overflows
            if name in ("raw2alpha", "raw2alpha_nonuni") and k == 1:             # alpha = 1 - pow(1 + e, -interval)
                np.testing.assert_allclose(bn[fin], an[fin], rtol=0, atol=1.2e-7, err_msg=key)
            else:                                                                # exp, gradients
                assert nc.ulp_diff(an[fin], bn[fin]).max() <= 4, key
    # the reference itself is unambiguous on these cases: its fma-contracted build produced the same bits
    import json
    fma = json.loads(bytes(z["report_json"]).decode())
    assert fma and all(v == 0 for v in fma.values())


def test_c_oracle_f64_pinned_on_reference_kernels(golden_dir):
    """oracle/ref_ops_f64.c -- the DOUBLE instantiation (the reference dispatches AT_DISPATCH_FLOATING_TYPES) -- against outputs of
    the reference's own kernels called with double tensors on an MI355X (tests/golden/gen_native_golden_f64.py ->
    native_ops_f64.npz): all 18 functions.  Bit-exact for everything but the raw2alpha family -- which shows that the restatement
    has the reference's float intermediates in the right places (a version that computed "everything in double" differs in
    every function that has one) -- and for that family exp / pow of glibc against the device libm: exp and the gradients within
    4 ulp of a double, alpha = 1 - pow(1 + e, -interval) within 1e-15 absolute."""
    import json
    import native_cases as nc
    gold, z = nc.load_golden(os.path.join(golden_dir, "native_ops_f64.npz"))
    orc = {nc.RU: ref_ops.render_utils_cuda, nc.TV: ref_ops.total_variation_cuda, nc.UB: ref_ops.ub360_utils_cuda,
           nc.AD: ref_ops.adam_upd_cuda}
    got = nc.run_all(orc, scale=1, chain_from=gold, dtype=torch.float64)
    n_f64 = 0
    for name in gold:
        assert len(got[name]) == len(gold[name]) > 0, name
        for k, (a, b) in enumerate(zip(gold[name], got[name])):
            key = "%s[%d]" % (name, k)
            assert a.shape == b.shape and a.dtype == b.dtype and a.dtype in (torch.float64, torch.int64, torch.bool), key
            n_f64 += a.dtype == torch.float64
            if name not in nc.TRANSCENDENTAL:
                assert np.array_equal(a.numpy(), b.numpy(), equal_nan=True), key
                continue
            an, bn = a.numpy(), b.numpy()
            assert np.array_equal(np.isfinite(an), np.isfinite(bn)), key
            fin = np.isfinite(an)
            assert np.array_equal(an[~fin], bn[~fin]), key
            if name in ("raw2alpha", "raw2alpha_nonuni") and k == 1:
                np.testing.assert_allclose(bn[fin], an[fin], rtol=0, atol=1e-15, err_msg=key)
            else:
                ai, bi = an[fin].view(np.int64), bn[fin].view(np.int64)        # (same sign: exp > 0; gradients compared where both finite)
                same_sign = np.signbit(an[fin]) == np.signbit(bn[fin])
                assert same_sign.all() and np.abs(ai - bi).max() <= 4, key
    assert n_f64 >= 25
    fma = json.loads(bytes(z["report_json"]).decode())
    assert fma and all(v == 0 for v in fma.values())      # the reference's two contraction builds agree on every output
    # the float32 entry points are refused doubles mixed with floats, and the ops without a reference counterpart stay fp32
    with pytest.raises(AssertionError):
        ref_ops.raw2alpha_nonuni(torch.zeros(4, dtype=torch.float64), 0.0, torch.zeros(4))
    with pytest.raises(AssertionError):
        ref_ops.segment_cumsum(torch.zeros(4, dtype=torch.float64), torch.zeros(4, dtype=torch.float64), torch.zeros(4, dtype=torch.int64))
