"""GPU parity of the drop-in ops (through the C ABI) against the CPU oracle on seeded inputs.

Bit-exact for everything that is integer / index work or pure IEEE add-mul-div-sqrt arithmetic
(both sides are compiled with FMA contraction off); exp/pow differ between glibc and the device
libm, so raw2alpha & co. are compared at a few-ulp tolerance stated next to each test.
"""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import ref_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from unboundednerfpytorch_amd import adam_upd_cuda, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    return render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda


def dev(*ts):
    return [t.cuda() if t is not None else None for t in ts]


def rays_case(seed, R, with_zero_dirs=True):
    o = torch.from_numpy(synth.normal(seed, R * 3, 0.0, 1.5).reshape(R, 3))
    d = torch.from_numpy(synth.normal(seed + 1, R * 3).reshape(R, 3))
    if with_zero_dirs and R >= 8:
        d[1, 0] = 0.0
        d[2] = torch.tensor([0.0, 0.0, 1.0])
        d[3, 2] = -0.0
    lo = torch.tensor([-1.0, -0.9, -1.1])
    hi = torch.tensor([1.0, 1.2, 0.8])
    return o, d, lo, hi


@pytest.mark.parametrize("R", [1, 63, 64, 4097])
def test_infer_and_sample_pts_bit_exact(mods, R):
    ru = mods[0]
    o, d, lo, hi = rays_case(100 + R, R)
    near, far, stepdist = 0.2, 1e9, 0.0173
    go, gd, glo, ghi = dev(o, d, lo, hi)
    t_ref = ref_ops.infer_t_minmax(o, d, lo, hi, near, far)
    t_gpu = ru.infer_t_minmax(go, gd, glo, ghi, near, far)
    for a, b in zip(t_ref, t_gpu):
        np.testing.assert_array_equal(a.numpy(), b.cpu().numpy())
    n_ref = ref_ops.infer_n_samples(d, t_ref[0], t_ref[1], stepdist)
    n_gpu = ru.infer_n_samples(gd, t_gpu[0], t_gpu[1], stepdist)
    np.testing.assert_array_equal(n_ref.numpy(), n_gpu.cpu().numpy())
    s_ref = ref_ops.infer_ray_start_dir(o, d, t_ref[0])
    s_gpu = ru.infer_ray_start_dir(go, gd, t_gpu[0])
    for a, b in zip(s_ref, s_gpu):
        np.testing.assert_array_equal(a.numpy(), b.cpu().numpy())
    full_ref = ref_ops.sample_pts_on_rays(o, d, lo, hi, near, far, stepdist)
    full_gpu = ru.sample_pts_on_rays(go, gd, glo, ghi, near, far, stepdist)
    assert len(full_gpu) == 7
    for a, b in zip(full_ref, full_gpu):
        assert a.dtype == b.dtype and a.shape == b.shape
        np.testing.assert_array_equal(a.numpy(), b.cpu().numpy())
    # ray_id monotone, step_id restarts at 0
    rid = full_gpu[2].cpu()
    assert torch.all(rid[1:] >= rid[:-1])


def test_sample_pts_many_rays_scan(mods):
    """> 1024*256 rays exercises the multi-block scan; totals must match the oracle exactly."""
    ru = mods[0]
    R = 300_001
    o, d, lo, hi = rays_case(7, R, with_zero_dirs=False)
    ref = ref_ops.sample_pts_on_rays(o, d, lo, hi, 0.0, 1e9, 0.21)
    gpu = ru.sample_pts_on_rays(*dev(o, d, lo, hi), 0.0, 1e9, 0.21)
    assert gpu[0].shape == ref[0].shape
    for a, b in zip(ref, gpu):
        np.testing.assert_array_equal(a.numpy(), b.cpu().numpy())


def test_empty_inputs(mods):
    ru, tv, ub, ad = mods
    z3 = torch.zeros(0, 3, device="cuda")
    lo, hi = torch.tensor([-1.0, -1, -1]).cuda(), torch.tensor([1.0, 1, 1]).cuda()
    out = ru.sample_pts_on_rays(z3, z3, lo, hi, 0.0, 1.0, 0.1)
    assert out[0].shape == (0, 3) and out[4].shape == (0,)
    e, a = ru.raw2alpha(torch.zeros(0, device="cuda"), 0.0, 0.5)
    assert e.numel() == 0 and a.numel() == 0
    w, T, last, i_s, i_e = ru.alpha2weight(torch.zeros(0, device="cuda"), torch.zeros(0, dtype=torch.int64, device="cuda"), 5)
    assert w.numel() == 0 and torch.all(last == 1) and torch.all(i_s == 0) and torch.all(i_e == 0)
    m = ru.maskcache_lookup(torch.ones(2, 2, 2, dtype=torch.bool, device="cuda"), z3, lo, hi)
    assert m.shape == (0,)


def test_sample_ndc_and_bg_bit_exact(mods):
    ru = mods[0]
    R, N = 513, 37
    o, d, lo, hi = rays_case(11, R)
    ref = ref_ops.sample_ndc_pts_on_rays(o, d, lo, hi, N)
    gpu = ru.sample_ndc_pts_on_rays(*dev(o, d, lo, hi), N)
    for a, b in zip(ref, gpu):
        np.testing.assert_array_equal(a.numpy(), b.cpu().numpy())
    t_max = torch.from_numpy(synth.uniform(12, R, 0.5, 3.0))
    refb = ref_ops.sample_bg_pts_on_rays(o, d, t_max, 0.3, N)
    gpub = ru.sample_bg_pts_on_rays(*dev(o, d, t_max), 0.3, N)
    np.testing.assert_array_equal(refb.numpy(), gpub.cpu().numpy())


def test_maskcache_lookup_bit_exact(mods):
    ru = mods[0]
    world = torch.from_numpy(synth.uniform(20, 9 * 7 * 5) > 0.5).reshape(9, 7, 5)
    n = 20_000
    xyz = torch.from_numpy(synth.uniform(21, n * 3, -1.6, 1.6).reshape(n, 3))
    xyz[:4] = torch.tensor([[-1.2, -1.2, -1.2], [1.2, 1.2, 1.2], [float("nan"), 0, 0], [1e30, 0, 0]])
    lo, hi = torch.full((3,), -1.2), torch.full((3,), 1.2)
    scale = (torch.tensor([9.0, 7.0, 5.0]) - 1) / (hi - lo)
    shift = -lo * scale
    # exact .5 positions exercise round-half-away-from-zero
    xyz[4:40, 0] = ((torch.arange(36) * 0.5 - 2.0) - shift[0]) / scale[0]
    ref = ref_ops.maskcache_lookup(world, xyz, scale, shift)
    gpu = ru.maskcache_lookup(*dev(world, xyz, scale, shift))
    np.testing.assert_array_equal(ref.numpy(), gpu.cpu().numpy())


def ulp_diff(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-2 ** 31) - ai, ai)
    bi = np.where(bi < 0, np.int64(-2 ** 31) - bi, bi)
    return np.abs(ai - bi)


def test_raw2alpha_and_backward(mods):
    """exp/pow come from different libms: tolerance 4 ulp on exp, and on alpha = 1 - pow(..) an absolute
    3e-7 (the subtraction from 1 amplifies pow's ulp when alpha is small)."""
    ru = mods[0]
    n = 100_003
    dens = torch.from_numpy(synth.normal(30, n, 4.0, 8.0))
    dens[:3] = torch.tensor([200.0, -200.0, 0.0])
    shift, interval = -9.21024, 0.5
    e_ref, a_ref = ref_ops.raw2alpha(dens, shift, interval)
    e_gpu, a_gpu = ru.raw2alpha(dens.cuda(), torch.tensor([shift]).cuda(), torch.tensor(interval))
    fin = torch.isfinite(e_ref).numpy()
    assert np.array_equal(fin, torch.isfinite(e_gpu).cpu().numpy())
    assert ulp_diff(e_ref.numpy()[fin], e_gpu.cpu().numpy()[fin]).max() <= 4
    np.testing.assert_allclose(a_gpu.cpu().numpy(), a_ref.numpy(), rtol=0, atol=3e-7)
    assert a_gpu[0] == 1 and a_gpu[1] == 0
    gb = torch.from_numpy(synth.normal(31, n))
    g_ref = ref_ops.raw2alpha_backward(e_ref, gb, interval)
    g_gpu = ru.raw2alpha_backward(e_ref.cuda(), gb.cuda(), interval)
    np.testing.assert_allclose(g_gpu.cpu().numpy(), g_ref.numpy(), rtol=2e-6, atol=1e-30)
    itv = torch.from_numpy(synth.uniform(32, n, 0.05, 1.5))
    e2_ref, a2_ref = ref_ops.raw2alpha_nonuni(dens, shift, itv)
    e2_gpu, a2_gpu = ru.raw2alpha_nonuni(dens.cuda(), shift, itv.cuda())
    np.testing.assert_allclose(a2_gpu.cpu().numpy(), a2_ref.numpy(), rtol=0, atol=3e-7)
    g2_ref = ref_ops.raw2alpha_nonuni_backward(e2_ref, gb, itv)
    g2_gpu = ru.raw2alpha_nonuni_backward(e2_ref.cuda(), gb.cuda(), itv.cuda())
    np.testing.assert_allclose(g2_gpu.cpu().numpy(), g2_ref.numpy(), rtol=2e-6, atol=1e-30)


def a2w_case(seed, n, R, amax):
    alpha = torch.from_numpy(synth.uniform(seed, n, 0.0, amax))
    rid = np.sort((synth.uniform(seed + 1, n) * R).astype(np.int64))
    rid[rid == 3] = 4  # leave some rays empty
    return alpha, torch.from_numpy(np.sort(rid))


@pytest.mark.parametrize("n,R,amax", [(1, 1, 0.5), (64, 1, 0.01), (65, 2, 0.2), (5000, 37, 0.05), (200_000, 513, 0.3),
                                      (30_000, 7, 0.002)])
def test_alpha2weight_bit_exact(mods, n, R, amax):
    ru = mods[0]
    alpha, rid = a2w_case(40 + n, n, R, amax)
    ref = ref_ops.alpha2weight(alpha, rid, R)
    gpu = ru.alpha2weight(alpha.cuda(), rid.cuda(), R)
    names = ["weight", "T", "alphainv_last", "i_start", "i_end"]
    for nm, a, b in zip(names, ref, gpu):
        np.testing.assert_array_equal(a.numpy(), b.cpu().numpy(), err_msg=nm)
    gw = torch.from_numpy(synth.normal(41, n))
    gl = torch.from_numpy(synth.normal(42, R))
    g_ref = ref_ops.alpha2weight_backward(alpha, *ref, R, gw, gl)
    g_gpu = ru.alpha2weight_backward(alpha.cuda(), *gpu, R, gw.cuda(), gl.cuda())
    np.testing.assert_array_equal(g_ref.numpy(), g_gpu.cpu().numpy())


def test_alpha2weight_properties_large(mods):
    """Size-independent properties at a frame-sized input: sum(w)+alphainv_last == 1 for unterminated rays,
    terminated rays stop below 1e-3, weights are 0 / T is 1 after the stop."""
    ru = mods[0]
    R, S = 8192, 668
    alpha = torch.from_numpy(synth.uniform(50, R * S, 0.0, 0.01)).cuda()
    alpha[: 100 * S] *= 50  # some rays terminate
    rid = torch.arange(R, device="cuda").repeat_interleave(S)
    w, T, last, i_s, i_e = ru.alpha2weight(alpha, rid, R)
    tot = torch.zeros(R, device="cuda").index_add_(0, rid, w) + last
    unterminated = last >= 1e-3
    assert torch.all((tot[unterminated] - 1).abs() < 5e-6)
    assert torch.all(i_e[unterminated] == i_s[unterminated] + S)
    assert (~unterminated).sum() > 10
    pos = torch.arange(R * S, device="cuda")
    after = pos >= i_e[rid]
    assert torch.all(w[after] == 0) and torch.all(T[after] == 1)


@pytest.mark.parametrize("shape", [(3, 2, 9, 6, 11), (2, 3, 5, 7, 12), (1, 1, 4, 4, 4), (7, 1, 20, 20, 20)])
@pytest.mark.parametrize("dense", [True, False])
def test_total_variation_bit_exact(mods, dense, shape):
    """(.., 11): scalar kernel; sz_k % 4 == 0: the float4 kernel -- both bit-identical to the oracle."""
    tv = mods[1]
    n = int(np.prod(shape))
    prm = torch.from_numpy(synth.normal(60, n, 0.0, 2.0).reshape(shape))
    g = synth.normal(61, n).reshape(shape)
    g[np.abs(g) < 0.7] = 0
    g_ref = torch.from_numpy(g.copy())
    g_gpu = torch.from_numpy(g.copy()).cuda()
    ref_ops.total_variation_add_grad(prm, g_ref, 0.3, 0.2, 0.1, dense)
    ret = tv.total_variation_add_grad(prm.cuda(), g_gpu, 0.3, 0.2, 0.1, dense)
    assert ret is None
    np.testing.assert_array_equal(g_ref.numpy(), g_gpu.cpu().numpy())
    if not dense:
        assert torch.all(g_gpu.cpu()[torch.from_numpy(g) == 0] == 0)


def test_cumdist_thres_bit_exact(mods):
    ub = mods[2]
    R, K = 301, 667
    dist = torch.from_numpy(synth.uniform(70, R * K, 0.0, 0.02).reshape(R, K))
    ref = ref_ops.cumdist_thres(dist, 0.0114)
    gpu = ub.cumdist_thres(dist.cuda(), 0.0114)
    np.testing.assert_array_equal(ref.numpy(), gpu.cpu().numpy())


@pytest.mark.parametrize("n", [5, 4096, 1_000_003])
def test_adam_family_bit_exact(mods, n):
    ad = mods[3]
    for mode in (0, 1, 2):
        p = torch.from_numpy(synth.normal(80, n))
        g = synth.normal(81, n)
        g[np.abs(g) < 1.0] = 0
        g = torch.from_numpy(g)
        m = torch.from_numpy(synth.normal(82, n, 0, 0.1))
        v = torch.from_numpy(synth.uniform(83, n, 0, 0.01))
        lr_ = torch.from_numpy(synth.uniform(84, n))
        ref = [p.clone(), m.clone(), v.clone()]
        gpu = [p.cuda(), m.cuda(), v.cuda()]
        args = (3, 0.9, 0.99, 0.1, 1e-8)
        if mode == 0:
            ref_ops.adam_upd(ref[0], g, ref[1], ref[2], *args)
            ad.adam_upd(gpu[0], g.cuda(), gpu[1], gpu[2], *args)
        elif mode == 1:
            ref_ops.masked_adam_upd(ref[0], g, ref[1], ref[2], *args)
            ad.masked_adam_upd(gpu[0], g.cuda(), gpu[1], gpu[2], *args)
        else:
            ref_ops.adam_upd_with_perlr(ref[0], g, ref[1], ref[2], lr_, *args)
            ad.adam_upd_with_perlr(gpu[0], g.cuda(), gpu[1], gpu[2], lr_.cuda(), *args)
        for a, b in zip(ref, gpu):
            np.testing.assert_array_equal(a.numpy(), b.cpu().numpy(), err_msg="mode %d" % mode)


def test_adam_unaligned_views(mods):
    """Views starting at a 4-byte offset take the scalar path; result identical to the vector path."""
    ad = mods[3]
    n = 1001
    vals = [torch.from_numpy(synth.normal(90 + i, n)) for i in range(4)]
    vals[3] = vals[3].abs() * 0.01
    vals[1][::3] = 0
    al = [v.cuda().clone() for v in vals]
    un = []
    for v in vals:
        buf = torch.empty(n + 1, device="cuda")
        buf[1:] = v.cuda()
        un.append(buf[1:])
    for t in un:
        assert t.data_ptr() % 16 != 0
    ad.masked_adam_upd(al[0], al[1], al[2], al[3], 2, 0.9, 0.99, 0.1, 1e-8)
    ad.masked_adam_upd(un[0], un[1], un[2], un[3], 2, 0.9, 0.99, 0.1, 1e-8)
    for a, b in zip(al, un):
        assert torch.equal(a, b)


def test_error_behaviour(mods):
    ru = mods[0]
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ru.raw2alpha(torch.zeros(4), 0.0, 0.5)
    x = torch.zeros(8, 3, device="cuda")[:, :2]
    with pytest.raises(RuntimeError, match="must be contiguous"):
        ru.infer_ray_start_dir(x, x, torch.zeros(8, device="cuda"))
    with pytest.raises(RuntimeError, match="float32 or float64"):      # the two types of the reference's AT_DISPATCH_FLOATING_TYPES
        ru.raw2alpha(torch.zeros(4, dtype=torch.float16, device="cuda"), 0.0, 0.5)
    e, a = ru.raw2alpha(torch.zeros(4, dtype=torch.float64, device="cuda"), 0.0, 0.5)      # (round 5: the double twins)
    assert e.dtype == a.dtype == torch.float64 and torch.equal(e, torch.ones_like(e))


def test_autograd_and_masked_adam_match_reference_golden(mods, golden_dir):
    """Host logic (autograd wiring, per-parameter optimizer dispatch) against vectors produced by the
    reference's own dvgo.Raw2Alpha / Alphas2Weights / masked_adam.MaskedAdam classes."""
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    from unboundednerfpytorch_amd.ops import Alphas2Weights, Raw2Alpha, Raw2Alpha_nonuni
    gold = np.load(os.path.join(golden_dir, "autograd_adam.npz"))
    n, R = 300, 17
    dens = torch.from_numpy(synth.normal(61, n, 5.0, 6.0)).cuda().requires_grad_(True)
    shift = torch.tensor([-9.21024]).cuda()
    alpha = Raw2Alpha.apply(dens, shift, 0.5)
    ray_id = torch.from_numpy(gold["a2w_ray_id"]).cuda()
    w, last = Alphas2Weights.apply(alpha, ray_id, R)
    gw = torch.from_numpy(synth.normal(63, n)).cuda()
    gl = torch.from_numpy(synth.normal(64, R)).cuda()
    (w * gw).sum().add((last * gl).sum()).backward()
    np.testing.assert_allclose(alpha.detach().cpu().numpy(), gold["a2w_alpha"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(w.detach().cpu().numpy(), gold["a2w_w"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(last.detach().cpu().numpy(), gold["a2w_last"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(dens.grad.cpu().numpy(), gold["a2w_grad_density"], rtol=2e-4, atol=1e-6)
    dens2 = torch.from_numpy(synth.normal(65, n, 5.0, 6.0)).cuda().requires_grad_(True)
    itv = torch.from_numpy(synth.uniform(66, n, 0.1, 1.0)).cuda()
    a2 = Raw2Alpha_nonuni.apply(dens2, shift, itv)
    (a2 * gw).sum().backward()
    np.testing.assert_allclose(a2.detach().cpu().numpy(), gold["nonuni_alpha"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(dens2.grad.cpu().numpy(), gold["nonuni_grad"], rtol=2e-5, atol=1e-7)

    shape = (1, 2, 4, 5, 6)
    p_grid = torch.nn.Parameter(torch.from_numpy(synth.normal(70, 240).reshape(shape)).cuda())
    p_dense = torch.nn.Parameter(torch.from_numpy(synth.normal(71, 33)).cuda())
    opt = MaskedAdam([{'params': [p_grid], 'lr': 0.1, 'skip_zero_grad': True},
                      {'params': [p_dense], 'lr': 1e-3, 'skip_zero_grad': False}])
    for step in range(3):
        g = synth.normal(80 + step, 240).reshape(shape)
        g[np.abs(g) < 0.8] = 0.0
        p_grid.grad = torch.from_numpy(g).cuda()
        p_dense.grad = torch.from_numpy(synth.normal(90 + step, 33)).cuda()
        if step == 2:
            opt.set_pervoxel_lr(torch.from_numpy(synth.uniform(95, 240, 0.0, 9.0).reshape(shape)).floor().cuda())
        opt.step()
    # Adam arithmetic is IEEE-only -> bit-exact against the reference run
    np.testing.assert_array_equal(p_grid.detach().cpu().numpy(), gold["adam_grid"])
    np.testing.assert_array_equal(p_dense.detach().cpu().numpy(), gold["adam_dense"])
    np.testing.assert_array_equal(opt.state[p_grid]['exp_avg'].cpu().numpy(), gold["adam_grid_m"])
    np.testing.assert_array_equal(opt.state[p_grid]['exp_avg_sq'].cpu().numpy(), gold["adam_grid_v"])


def test_sharded_masked_adam_degenerates_to_masked_adam_on_one_gpu():
    """Without a process group ShardedMaskedAdam must step exactly like MaskedAdam on the HIP kernels (the sharded
    path itself is covered by the 2-rank gloo test in test_host_logic.py)."""
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    from unboundednerfpytorch_amd.sharded_adam import ShardedMaskedAdam
    shape = (7, 3, 9, 8, 8)
    n = int(np.prod(shape))
    base = torch.from_numpy(synth.normal(61, n).reshape(shape)).cuda()
    a, b = torch.nn.Parameter(base.clone()), torch.nn.Parameter(base.clone())
    oa = MaskedAdam([{'params': [a], 'lr': 0.1, 'skip_zero_grad': True}])
    ob = ShardedMaskedAdam([{'params': [b], 'lr': 0.1, 'skip_zero_grad': True}])
    for step in range(3):
        g = torch.from_numpy(synth.normal(62 + step, n).reshape(shape))
        g = torch.where(torch.from_numpy(synth.uniform(70 + step, n).reshape(shape)) < 0.3, g, torch.zeros_like(g)).cuda()
        a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    assert torch.equal(a.data, b.data)
    assert torch.equal(oa.state[a]['exp_avg_sq'], ob.state[b]['exp_avg_sq'])


def test_masked_adam_keeps_the_gradient_like_the_reference_unless_recycling_is_requested():
    """The reference's MaskedAdam leaves `.grad` untouched by step() (masked_adam.py:43-75).  The drop-in class does the same
    by default; `recycle_grads=True` (this package's training loop) re-zeroes the buffer inside the update kernel, parks it
    for the next backward and sets `.grad = None`.  With the pool switched off nothing is dropped, whatever the flag says
    (a refused `give` must not free a buffer a side-stream kernel is still using).  Same parameter update in
    all three cases, with and without the TV term on a second stream."""
    from unboundednerfpytorch_amd import _gradpool
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    shape = (3, 4, 8, 8, 8)
    n = int(np.prod(shape))
    base = torch.from_numpy(synth.normal(161, n).reshape(shape)).cuda()
    g0 = torch.from_numpy(synth.normal(162, n).reshape(shape))
    g0 = torch.where(torch.from_numpy(synth.uniform(163, n).reshape(shape)) < 0.3, g0, torch.zeros_like(g0)).cuda()
    for tv in (None, "main", "side"):
        res = []
        for recycle, pool in ((False, True), (True, True), (True, False)):
            _gradpool.clear()
            _gradpool.enabled = pool
            try:
                p = torch.nn.Parameter(base.clone())
                opt = MaskedAdam([{'params': [p], 'lr': 0.1, 'skip_zero_grad': True}], recycle_grads=recycle)
                p.grad = g0.clone()
                if tv is None:
                    opt.step()
                else:
                    opt.step(tv_terms={p: (1e-3, True, None)}, overlap=[p] if tv == "side" else None)
                torch.cuda.synchronize()
                from unboundednerfpytorch_amd._lib import wait_pending
                wait_pending(p)
                if recycle and pool:
                    assert p.grad is None
                    buf = _gradpool._POOL[id(p)][1]
                    assert buf is not None and float(buf.abs().max()) == 0.0
                else:
                    assert p.grad is not None                       # never dropped
                    if not recycle:
                        assert tv is not None or torch.equal(p.grad, g0)      # untouched (the TV term is added in place, as in the reference)
                res.append(p.detach().clone())
            finally:
                _gradpool.enabled = True
                _gradpool.clear()
        assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2]), tv


def test_segment_cumsum_and_distortion_loss(mods, golden_dir):
    """segment_cumsum (HIP, wave per ray, serial fp32 chains) is bit-exact against the oracle op; DistortionLoss on
    top of it reproduces the loss and gradient of the reference's own class (tests/golden/distortion.npz)."""
    from unboundednerfpytorch_amd import ub360_utils_cuda as ub
    from unboundednerfpytorch_amd.ops import DistortionLoss, distortion_loss
    gold = np.load(os.path.join(golden_dir, "distortion.npz"))
    w, s, ray_id, n_max = synth.distortion_inputs()
    wd, sd, rd = torch.from_numpy(w).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(ray_id).cuda()
    got = ub.segment_cumsum(wd, sd, rd)
    for g, key in zip(got, ("w_prefix", "w_total", "ws_prefix", "ws_total")):
        assert np.array_equal(g.cpu().numpy(), gold[key]), key
    # a long random case against the oracle op, incl. empty rays at both ends and an explicit n_rays
    n_rays = 300
    counts = np.floor(synth.uniform(41, n_rays, 0.0, 150.0)).astype(np.int64)
    counts[[0, 7, n_rays - 1]] = 0
    rid = np.concatenate([np.full(c, r, dtype=np.int64) for r, c in enumerate(counts)])
    ww = synth.uniform(42, rid.shape[0], 0.0, 1.0)
    ss = synth.uniform(43, rid.shape[0], 0.0, 1.0)
    ref = ref_ops.segment_cumsum(torch.from_numpy(ww), torch.from_numpy(ss), torch.from_numpy(rid), n_rays)
    dev = ub.segment_cumsum(torch.from_numpy(ww).cuda(), torch.from_numpy(ss).cuda(), torch.from_numpy(rid).cuda(), n_rays)
    for a, b in zip(ref, dev):
        assert torch.equal(a, b.cpu())
    wt = wd.clone().requires_grad_(True)
    loss = distortion_loss(wt, sd, n_max, rd)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(gold["loss"]), rtol=2e-6)
    np.testing.assert_allclose(wt.grad.cpu().numpy(), gold["grad"], rtol=2e-6, atol=1e-7)
    assert DistortionLoss.apply is not None


@pytest.mark.parametrize("shape", [(3, 2, 9, 6, 12), (1, 1, 4, 4, 4), (7, 1, 20, 20, 20), (2, 3, 5, 7, 11)])
@pytest.mark.parametrize("skip_zero", [True, False])
def test_fused_dense_tv_adam_is_bit_identical_to_the_two_reference_calls(mods, shape, skip_zero):
    """adam_upd_cuda.tv_adam_dense (one pass, out-of-place parameter) == total_variation_add_grad(dense) followed by
    masked_adam_upd / adam_upd, bit for bit; shapes whose last dimension is not a multiple of 4 report 'not supported'
    (the optimizer then uses the two calls).  Also through MaskedAdam.step(tv_terms=...), which swaps the buffers."""
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    tv, ad = mods[1], mods[3]
    n = int(np.prod(shape))
    p = torch.from_numpy(synth.normal(60, n, 0.0, 2.0).reshape(shape)).cuda()
    g = synth.normal(61, n).reshape(shape)
    g[np.abs(g) < 0.7] = 0
    g = torch.from_numpy(g).cuda()
    m = torch.from_numpy(synth.normal(62, n, 0, 0.1).reshape(shape)).cuda()
    v = torch.from_numpy(synth.uniform(63, n, 0, 0.01).reshape(shape)).cuda()
    w, args = 0.37, (5, 0.9, 0.99, 0.1, 1e-8)
    p_ref, g_ref, m_ref, v_ref = p.clone(), g.clone(), m.clone(), v.clone()
    tv.total_variation_add_grad(p_ref, g_ref, w, w, w, True)
    (ad.masked_adam_upd if skip_zero else ad.adam_upd)(p_ref, g_ref, m_ref, v_ref, *args)
    out, m2, v2 = torch.empty_like(p), m.clone(), v.clone()
    ok = ad.tv_adam_dense(p, out, g, m2, v2, w, w, w, *args, skip_zero)
    assert ok == (shape[-1] % 4 == 0)
    if ok:
        assert torch.equal(out, p_ref) and torch.equal(m2, m_ref) and torch.equal(v2, v_ref)
        assert torch.equal(g, torch.from_numpy(np.where(np.abs(synth.normal(61, n).reshape(shape)) < 0.7, 0, synth.normal(61, n).reshape(shape))).cuda())
    # optimizer level: three steps with a TV term, fused vs the hook-free reference sequence
    pa, pb = torch.nn.Parameter(p.clone()), torch.nn.Parameter(p.clone())
    oa = MaskedAdam([{'params': [pa], 'lr': 0.1, 'skip_zero_grad': skip_zero}])
    ob = MaskedAdam([{'params': [pb], 'lr': 0.1, 'skip_zero_grad': skip_zero}])
    for step in range(3):
        gs = torch.from_numpy(synth.normal(70 + step, n).reshape(shape)).cuda()
        pa.grad, pb.grad = gs.clone(), gs.clone()
        oa.step(tv_terms={pa: (w, True, None)})
        tv.total_variation_add_grad(pb, pb.grad, w, w, w, True)
        ob.step()
    assert torch.equal(pa.data, pb.data) and torch.equal(oa.state[pa]['exp_avg_sq'], ob.state[pb]['exp_avg_sq'])


@pytest.mark.parametrize("channels_last", [False, True])
def test_fused_tv_adam_rezero_grad_returns_the_gradient_buffer_all_zero(channels_last):
    from unboundednerfpytorch_amd import adam_upd_cuda
    shape = (3, 4, 9, 10, 12)
    n = int(np.prod(shape))
    fmt = torch.channels_last_3d if channels_last else torch.contiguous_format
    mk = lambda seed, lo=None: torch.from_numpy(synth.normal(seed, n).reshape(shape)).cuda().contiguous(memory_format=fmt)
    p, g, m, v = mk(1), mk(2), mk(3) * 0.1, mk(4).abs() * 0.01
    g[g.abs() < 1.0] = 0            # sparse, like a scattered gradient
    g[0, 0, 0, 0, 0] = float("nan")
    outs = []
    for rezero in (False, True):
        gi, mi, vi, out = g.clone(memory_format=torch.preserve_format), m.clone(memory_format=torch.preserve_format), \
            v.clone(memory_format=torch.preserve_format), torch.empty_like(p)
        assert adam_upd_cuda.tv_adam_dense(p, out, gi, mi, vi, 0.2, 0.2, 0.2, 5, 0.9, 0.99, 0.1, 1e-8, True, rezero_grad=rezero)
        outs.append((out, mi, vi))
        if rezero:
            assert not bool(gi.any())
        else:
            assert torch.equal(torch.nan_to_num(gi), torch.nan_to_num(g))
    for a, b in zip(*outs):
        assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))


@pytest.mark.parametrize("n", [4096 * 3 + 2, 1 << 16])
def test_masked_adam_rezero_grad_returns_the_gradient_buffer_all_zero(n):
    from unboundednerfpytorch_amd import adam_upd_cuda
    mk = lambda seed: torch.from_numpy(synth.normal(seed, n)).cuda()
    p, g, m, v = mk(1), mk(2), mk(3) * 0.1, mk(4).abs() * 0.01
    g[g.abs() < 1.2] = 0
    args = (7, 0.9, 0.99, 0.1, 1e-8)
    pa, ga, ma, va = p.clone(), g.clone(), m.clone(), v.clone()
    pb, gb, mb, vb = p.clone(), g.clone(), m.clone(), v.clone()
    adam_upd_cuda.masked_adam_upd(pa, ga, ma, va, *args)
    adam_upd_cuda.masked_adam_upd_rezero(pb, gb, mb, vb, *args)
    assert torch.equal(ga, g) and not bool(gb.any())
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)


@pytest.mark.gpu
@pytest.mark.parametrize("M,C,E,W", [(5000, 12, 27, 128), (33, 12, 27, 128), (1, 3, 15, 128), (4097, 15, 27, 128), (2500, 12, 51, 128),
                                     (70000, 12, 27, 128), (3000, 9, 27, 64), (777, 9, 27, 40)])
def test_fused_rgbnet_matches_torch_linear_layers(M, C, E, W):
    """ops.FusedRgbnet (fp32-MFMA kernels, csrc/ugrid_train_mlp.hip) vs the same three nn.Linear layers through torch: logits and
    every gradient (k0 features, weights, biases); fp32 products on both sides, so only the summation order differs"""
    from unboundednerfpytorch_amd import ops
    torch.manual_seed(M + C)
    net = torch.nn.Sequential(torch.nn.Linear(C + E, W), torch.nn.ReLU(inplace=True),
                              torch.nn.Sequential(torch.nn.Linear(W, W), torch.nn.ReLU(inplace=True)), torch.nn.Linear(W, 3)).cuda()
    with torch.no_grad():
        net[3].bias.normal_(0, 0.1)
    lin = ops.rgbnet_linears(net)
    assert lin is not None and [l.in_features for l in lin] == [C + E, W, W]
    assert ops.rgbnet_linears(torch.nn.Sequential(torch.nn.Linear(C + E, 64), torch.nn.ReLU(), torch.nn.Linear(64, 3))) is None
    k0 = torch.randn(M, C, device="cuda", requires_grad=True)
    emb = torch.randn(M, E, device="cuda")
    go = torch.randn(M, 3, device="cuda")
    ref = net(torch.cat([k0, emb], -1))
    ref.backward(go)
    want = [k0.grad.clone()] + [p.grad.clone() for l in lin for p in (l.weight, l.bias)]
    k0.grad = None
    net.zero_grad(set_to_none=True)
    out = ops.FusedRgbnet.apply(k0, emb, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias)
    out.backward(go)
    got = [k0.grad] + [p.grad for l in lin for p in (l.weight, l.bias)]
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-5)
    # a pre-activation within rounding of zero can land on either side of the ReLU in the two summation orders: that sample's
    # gradient then differs by a whole term (1 of 70 000 x 256 activations in the largest case).  Rows: at most M / 20000 + 1
    # such samples; weight gradients: one sample's contribution of slack on top of the rounding-level bound.
    row_err = (got[0] - want[0]).abs().amax(dim=1)
    flips = int((row_err > 3e-5 * float(want[0].abs().max()) + 1e-6).sum())
    assert flips <= M // 20000 + 1, flips
    for n, a, b in zip(["w0", "b0", "w1", "b1", "w2", "b2"], got[1:], want[1:]):
        scale = float(b.abs().max()) + 1e-12
        tol = (3e-5 if flips == 0 else 1e-2) * scale + 1e-6
        assert float((a - b).abs().max()) <= tol, (n, float((a - b).abs().max()), scale, flips)


@pytest.mark.gpu
@pytest.mark.parametrize("M,C,E,W", [(70000, 12, 27, 128), (4097, 15, 27, 128), (3000, 9, 27, 64), (777, 9, 27, 40), (31, 12, 51, 128)])
def test_fused_rgbnet_bf16x3_kernels_against_the_fp32_mfma_kernels(M, C, E, W):
    """ugrid_tune('train_mlp', 1) (k_lin_b3 / k_wgrad_b3: every fp32 operand split into three bf16 parts, the six part products above
    2^-24, fp32 accumulation -- the default since round 6) against ugrid_tune('train_mlp', 0) (v_mfma_f32_32x32x2_f32: exact fp32
    products) on the same inputs: logits and every gradient to 1e-6 of the tensor's largest magnitude -- two orders below the suite's
    1e-4 bound for fp32 gradients -- apart from the samples whose pre-activation sits within rounding of zero and lands on the other
    side of the ReLU (at most M / 20000 + 1 rows; they carry a whole term into the weight gradients).  Inputs at trained-like scales
    AND at 1e-6 / 1e+4 (the gradient's and a saturated feature's range: bf16 keeps fp32's exponent, no range guard)."""
    from unboundednerfpytorch_amd import ops, fourier_render as fr
    torch.manual_seed(M + W)
    net = torch.nn.Sequential(torch.nn.Linear(C + E, W), torch.nn.ReLU(inplace=True),
                              torch.nn.Sequential(torch.nn.Linear(W, W), torch.nn.ReLU(inplace=True)), torch.nn.Linear(W, 3)).cuda()
    lin = ops.rgbnet_linears(net)
    par = [p for l in lin for p in (l.weight, l.bias)]
    for scale_in, scale_go in ((1.0, 1.0), (1e4, 1e-6)):
        k0 = (torch.randn(M, C, device="cuda") * scale_in).requires_grad_(True)
        emb = torch.randn(M, E, device="cuda")
        go = torch.randn(M, 3, device="cuda") * scale_go
        res = []
        try:
            for mode in (0, 1):
                fr.tune("train_mlp", mode)
                k0.grad = None
                net.zero_grad(set_to_none=True)
                out = ops.FusedRgbnet.apply(k0, emb, *par)
                out.backward(go)
                res.append([out.detach().clone(), k0.grad.clone()] + [p.grad.clone() for p in par])
        finally:
            fr.tune("train_mlp", 1)
        a, b = res
        assert not torch.equal(a[0], b[0]) or M < 64            # (the two arithmetics are different roundings: the knob did something)
        assert float((a[0] - b[0]).abs().max()) <= 1e-6 * float(a[0].abs().max()) + 1e-30
        row_err = (a[1] - b[1]).abs().amax(dim=1)
        flips = int((row_err > 1e-6 * float(a[1].abs().max()) + 1e-30).sum())
        assert flips <= M // 20000 + 1, flips
        for n, x, y in zip(["w0", "b0", "w1", "b1", "w2", "b2"], a[2:], b[2:]):
            scale = float(x.abs().max()) + 1e-30
            tol = (1e-6 if flips == 0 else 1e-2) * scale
            assert float((x - y).abs().max()) <= tol, (n, float((x - y).abs().max()), scale, flips, scale_in)


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,C,pe", [(8192, 100000, 12, 4), (77, 1000, 9, 4), (5, 0, 12, 4), (300, 4097, 3, 8), (64, 500, 0, 4),
                                      (1, 1, 12, 0), (900, None, 12, 4), (900, 1000, 12, 4)])
def test_rgbnet_features_equals_the_torch_chain(N, M, C, pe):
    """ops.rgbnet_features (one kernel) vs the reference's chain (FourierGrid_model.py:631-635): the k0 and viewdir columns are
    copies (bit-equal); the sin / cos columns are sinf / cosf of the same fp32 product, within 1 ulp-of-one of torch's device
    sin / cos (the two are built from different releases of the device math library, so the last bit may differ).  M = 0, no
    k0 columns (embedding rows only), pe = 0 and ray_id = None (one row per ray) included; then the same rows fed to
    FusedRgbnet as a ViewRows give the logits and gradients of the explicit-embedding call bit for bit.  Cases with at least two
    samples per ray take the per-ray embedding + gather (two launches), the others the one-pass kernel: the same values."""
    from unboundednerfpytorch_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N + 3 * pe)
    viewdirs = torch.nn.functional.normalize(torch.randn(N, 3, device="cuda", generator=g), dim=-1)
    viewfreq = torch.tensor([float(2 ** i) for i in range(pe)], device="cuda")
    if M is None:
        ray_id, rows = None, N
    else:
        ray_id = torch.randint(0, N, (M,), device="cuda", generator=g).sort().values
        rows = M
    k0 = torch.randn(rows, C, device="cuda", generator=g) if C else None
    got = ops.rgbnet_features(k0, viewdirs, viewfreq, ray_id)
    e = (viewdirs.unsqueeze(-1) * viewfreq).flatten(-2)
    emb = torch.cat([viewdirs, e.sin(), e.cos()], -1)
    emb = emb if ray_id is None else emb[ray_id]
    want = emb if k0 is None else torch.cat([k0, emb], -1)
    assert got.shape == want.shape == (rows, C + 3 + 6 * pe)
    assert torch.equal(got[:, :C + 3], want[:, :C + 3])
    if rows and pe:
        assert float((got - want).abs().max()) <= 1.2e-7, float((got - want).abs().max())
    # [..., 3] view directions of an image (H, W, 3) flatten to rays
    if M is not None and N % 4 == 0 and rows:
        again = ops.rgbnet_features(k0, viewdirs.reshape(4, N // 4, 3), viewfreq, ray_id)
        assert torch.equal(again, got)
    if C and rows and pe == 4:
        net = torch.nn.Sequential(torch.nn.Linear(C + 27, 128), torch.nn.ReLU(inplace=True),
                                  torch.nn.Sequential(torch.nn.Linear(128, 128), torch.nn.ReLU(inplace=True)), torch.nn.Linear(128, 3)).cuda()
        lin = ops.rgbnet_linears(net)
        par = [p for l in lin for p in (l.weight, l.bias)]
        go = torch.randn(rows, 3, device="cuda", generator=g)
        res = []
        for second in (got[:, C:].contiguous(), ops.ViewRows(viewdirs, viewfreq, ray_id)):
            k = k0.clone().requires_grad_(True)
            net.zero_grad(set_to_none=True)
            out = ops.FusedRgbnet.apply(k, second, *par)
            out.backward(go)
            res.append([out.detach().clone(), k.grad.clone()] + [p.grad.clone() for p in par])
        for a, b in zip(*res):
            assert torch.equal(a, b)
    with pytest.raises(TypeError):
        ops.rgbnet_features(k0, viewdirs, viewfreq, torch.zeros(rows, dtype=torch.int32, device="cuda"))
    if C:
        with pytest.raises(ValueError):
            ops.rgbnet_features(torch.zeros(rows + 1, C, device="cuda"), viewdirs, viewfreq, ray_id)


@pytest.mark.parametrize("shape", [(1, 12, 8, 128, 96), (2, 4, 5, 300, 128), (3, 12, 24, 40, 40)])
@pytest.mark.parametrize("skip_zero", [True, False])
def test_slab_ordered_dense_tv_adam_is_bit_identical(shape, skip_zero):
    """Round 5: the fused dense TV + Adam pass on channel-last grids visits the array in slabs of j-rows, i fastest over the slabs'
    planes (k_tv_cl_slab, ugrid_tune tv_xcd = 3, the default), so that the i-1 / i+1 neighbour planes are found in L2 instead of
    being fetched from memory three times.  Same loads, same expression per element: the result must equal the linear kernel's
    (tv_xcd = 2) and the reference's two calls (total_variation_add_grad(dense) then (masked_)adam_upd) bit for bit -- three slabs
    with a ragged last one (128 = 56 + 56 + 16 rows; 300 = 128 + 128 + 44, two levels), and a grid whose planes are too small for
    the slab order (falls back to the linear kernel: same bits again)."""
    from unboundednerfpytorch_amd import adam_upd_cuda as ad, total_variation_cuda as tv
    from unboundednerfpytorch_amd.fourier_render import tune
    g = torch.Generator(device="cuda").manual_seed(17)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last_3d)
    p0 = cl(torch.randn(shape, device="cuda", generator=g))
    gr = cl(torch.randn(shape, device="cuda", generator=g) * (torch.rand(shape, device="cuda", generator=g) < 0.1))
    m0 = cl(torch.randn(shape, device="cuda", generator=g) * 0.01)
    v0 = cl(torch.rand(shape, device="cuda", generator=g) * 0.01)
    args = (5, 0.9, 0.99, 0.1, 1e-8)
    outs = {}
    try:
        for mode in (2, 3):
            tune("tv_xcd", mode)
            m, v = m0.clone(memory_format=torch.preserve_format), v0.clone(memory_format=torch.preserve_format)
            out = torch.empty_like(p0, memory_format=torch.preserve_format)
            assert ad.tv_adam_dense(p0, out, gr, m, v, 0.3, 0.3, 0.3, *args, skip_zero)
            outs[mode] = (out, m, v)
    finally:
        tune("tv_xcd", 3)
    for a, b in zip(outs[2], outs[3]):
        assert torch.equal(a, b)
    # ... and the reference's two calls
    pr, gr2 = p0.clone(memory_format=torch.preserve_format), gr.clone(memory_format=torch.preserve_format)
    mr, vr = m0.clone(memory_format=torch.preserve_format), v0.clone(memory_format=torch.preserve_format)
    tv.total_variation_add_grad(pr, gr2, 0.3, 0.3, 0.3, True)
    (ad.masked_adam_upd if skip_zero else ad.adam_upd)(pr, gr2, mr, vr, *args)
    assert torch.equal(outs[3][0], pr) and torch.equal(outs[3][1], mr) and torch.equal(outs[3][2], vr)
    assert not torch.equal(pr, p0)
