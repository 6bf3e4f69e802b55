"""Touched-line bitmap of the recycled grid gradient (include/ugrid_hip.h: ugrid_*_touch; _gradpool.py): the lookup backward
marks the 256-byte lines it adds to, and the masked TV / masked Adam / fused dense TV + Adam passes visit only those.  Given
the same gradient the passes must produce the SAME BITS as the scanning kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(P=3, C=12, X=21, Y=18, Z=23, n=3000, seed=0):
    from unboundednerfpytorch_amd import _lib
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    grid = torch.randn(P, C, X, Y, Z, device="cuda", generator=g).contiguous(memory_format=torch.channels_last_3d)
    # points clustered in a corner of the box: most of the gradient stays untouched
    pts = (torch.rand(n, 3, device="cuda", generator=g) * 0.5 - 0.9).contiguous()
    pts[:7] = torch.tensor([0.999, -0.999, 0.5], device="cuda")          # border cells
    pts[7:9] = 1.7                                                         # outside: zero padding, no gradient
    go = torch.randn(n, C, device="cuda", generator=g)
    go[100:200] = 0                                                        # exact zeros stay exact zeros
    go[300:400, 1:] = 0                                                    # only the first channel of a record
    go[400:500, :-1] = 0                                                   # only the last
    lo = torch.tensor([-1.0, -1.0, -1.0], device="cuda")
    hi = torch.tensor([1.0, 1.0, 1.0], device="cuda")
    return _lib, grid, pts, go, lo, hi


def _backward(_lib, grid, pts, go, lo, hi, F, touch):
    L = _lib.load()
    P, C, X, Y, Z = grid.shape
    gg = torch.zeros_like(grid, memory_format=torch.preserve_format)
    st = torch.cuda.current_stream().cuda_stream
    if touch is None:
        _lib.check(L.ugrid_grid_query_backward_cl(go.data_ptr(), P, C, X, Y, Z, pts.data_ptr(), lo.data_ptr(), hi.data_ptr(), F,
                                                  pts.shape[0], gg.data_ptr(), st), "bwd")
    else:
        _lib.check(L.ugrid_grid_query_backward_cl_touch(go.data_ptr(), P, C, X, Y, Z, pts.data_ptr(), lo.data_ptr(), hi.data_ptr(), F,
                                                        pts.shape[0], gg.data_ptr(), touch.data_ptr(), st), "bwd touch")
    return gg


def _flat(t):
    return t.permute(0, 2, 3, 4, 1).reshape(-1)


def _bits(touch, n_lines):
    w = touch.to(torch.int64) & 0xFFFFFFFF
    b = ((w[:, None] >> torch.arange(32, device=touch.device)[None, :]) & 1).reshape(-1)
    return b[:n_lines].bool()


def test_backward_marks_every_line_it_writes():
    _lib, grid, pts, go, lo, hi = _setup()
    L = _lib.load()
    N = grid.numel()
    words = int(L.ugrid_touch_words(N))
    assert words == ((N + 63) // 64 + 31) // 32
    touch = torch.zeros(words, dtype=torch.int32, device="cuda")
    a = _backward(_lib, grid, pts, go, lo, hi, 1, None)
    b = _backward(_lib, grid, pts, go, lo, hi, 1, touch)
    torch.testing.assert_close(_flat(a), _flat(b), rtol=1e-4, atol=1e-5)      # atomics: same sums up to the order
    n_lines = (N + 63) // 64
    fb = _flat(b)
    pad = torch.zeros(n_lines * 64 - N, device="cuda")
    line_nz = (torch.cat([fb, pad]).view(n_lines, 64) != 0).any(dim=1)
    bits = _bits(touch, n_lines)
    assert bool((bits | ~line_nz).all()), "a written line is not marked"
    frac = float(bits.float().mean())
    assert 0.0 < frac < 0.5, frac                                              # and the bitmap is selective
    assert float(line_nz.float().mean()) > 0.5 * frac                          # most marked lines do hold a gradient


@pytest.mark.parametrize("with_tv", [True, False])
def test_masked_tv_and_adam_on_marked_lines_equal_the_scanning_kernels(with_tv):
    from unboundednerfpytorch_amd import adam_upd_cuda, total_variation_cuda
    _lib, grid, pts, go, lo, hi = _setup(seed=1)
    L = _lib.load()
    touch = torch.zeros(int(L.ugrid_touch_words(grid.numel())), dtype=torch.int32, device="cuda")
    g_ref = _backward(_lib, grid, pts, go, lo, hi, 1, touch)
    assert int((touch != 0).sum()) > 0
    outs = []
    for use_touch in (False, True):
        p = grid.clone(memory_format=torch.preserve_format)
        g = g_ref.clone(memory_format=torch.preserve_format)
        m = torch.full_like(p, 0.01, memory_format=torch.preserve_format)
        v = torch.full_like(p, 0.002, memory_format=torch.preserve_format)
        t = touch.clone()
        if with_tv:
            if use_touch:
                total_variation_cuda.total_variation_add_grad_touched(p, g, 1e-3, 1e-3, 1e-3, t)
            else:
                total_variation_cuda.total_variation_add_grad(p, g, 1e-3, 1e-3, 1e-3, False)
        g_after_tv = g.clone(memory_format=torch.preserve_format)
        adam_upd_cuda.masked_adam_upd_rezero(p, g, m, v, 7, 0.9, 0.99, 0.1, 1e-8, **({"touch": t} if use_touch else {}))
        assert float(g.abs().max()) == 0.0                                     # the buffer goes back to the pool all zero
        if use_touch:
            assert int((t != 0).sum()) == 0                                    # and the bitmap cleared
        outs.append((g_after_tv, p, m, v))
    for a, b, name in zip(outs[0], outs[1], ("grad after TV", "param", "exp_avg", "exp_avg_sq")):
        assert torch.equal(a, b), name
    assert not torch.equal(outs[0][1], grid)                                   # something was updated
    untouched = _flat(g_ref) == 0
    assert torch.equal(_flat(outs[1][1])[untouched], _flat(grid)[untouched])   # masked: zero-gradient entries keep their value


@pytest.mark.parametrize("skip_zero_grad", [True, False])
def test_fused_dense_pass_with_bitmap_equals_without(skip_zero_grad):
    from unboundednerfpytorch_amd import adam_upd_cuda
    _lib, grid, pts, go, lo, hi = _setup(seed=2)
    L = _lib.load()
    touch = torch.zeros(int(L.ugrid_touch_words(grid.numel())), dtype=torch.int32, device="cuda")
    g_ref = _backward(_lib, grid, pts, go, lo, hi, 1, touch)
    outs = []
    for use_touch in (False, True):
        p = grid.clone(memory_format=torch.preserve_format)
        po = torch.empty_like(p, memory_format=torch.preserve_format)
        g = g_ref.clone(memory_format=torch.preserve_format)
        m = torch.full_like(p, 0.01, memory_format=torch.preserve_format)
        v = torch.full_like(p, 0.002, memory_format=torch.preserve_format)
        t = touch.clone()
        ok = adam_upd_cuda.tv_adam_dense(p, po, g, m, v, 2e-3, 2e-3, 2e-3, 3, 0.9, 0.99, 0.1, 1e-8, skip_zero_grad, rezero_grad=True,
                                         **({"touch": t} if use_touch else {}))
        assert ok
        assert float(g.abs().max()) == 0.0
        if use_touch:
            assert int((t != 0).sum()) == 0
        outs.append((po, m, v))
    for a, b, name in zip(outs[0], outs[1], ("param_out", "exp_avg", "exp_avg_sq")):
        assert torch.equal(a, b), name


def test_optimizer_uses_the_bitmap_and_recycles_it():
    """GridQuery.backward -> MaskedAdam(recycle_grads).step with a masked TV term: the bitmap is bound to the recycled buffer,
    served to the optimizer, cleared by it, and the same tensors come back on the next step; results equal a run with the
    bitmaps switched off up to the atomics' summation order"""
    from unboundednerfpytorch_amd import _gradpool
    from unboundednerfpytorch_amd.grid import GridQuery
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    _lib, grid0, pts, go, lo, hi = _setup(seed=3)
    res = {}
    for on in (True, False):
        _gradpool.clear()
        _gradpool.touch_enabled = on
        try:
            p = torch.nn.Parameter(grid0.clone(memory_format=torch.preserve_format))
            opt = MaskedAdam([{"params": [p], "lr": 0.1, "skip_zero_grad": True}], recycle_grads=True)
            seen = []
            for it in range(3):
                _gradpool.certify([p])          # what train_step.train_iteration does for the model's own loss graph
                out = GridQuery.apply(p, pts, lo, hi, 1)
                (out * go).sum().backward()
                t = _gradpool.touch_of(p, p.grad)
                assert (t is not None) == on
                if on:
                    assert int((t != 0).sum()) > 0
                    seen.append((p.grad.data_ptr(), t.data_ptr()))
                opt.step(tv_terms={p: (1e-3, it == 0, None)})                   # dense fused pass first, then masked passes
                assert p.grad is None                                           # parked
                if on:
                    assert int((t != 0).sum()) == 0
            if on:
                assert len(set(seen)) == 1, seen                                # one buffer, one bitmap, recycled
            res[on] = p.detach().clone()
        finally:
            _gradpool.touch_enabled = True
            _gradpool.clear()
    _gradpool.clear()
    # (the scatter's atomics make a gradient that cancels to ~0 land on either side of zero: Adam's first steps then differ by
    #  up to 2 lr on that element -- allow a 1e-5 fraction of such elements, everything else agrees to rounding)
    diff = (res[True] - res[False]).abs()
    assert int((diff > 1e-4).sum()) <= max(2, int(1e-5 * diff.numel())), (float(diff.max()), int((diff > 1e-4).sum()))


def test_a_second_gradient_producer_never_meets_a_bitmap():
    """AccumulateGrad adds a second producer's gradient IN PLACE into the first-arrived buffer, so
    `p.pow(2).sum() + GridQuery(p)` leaves .grad at the lookup's buffer address holding dense values its bitmap never saw.
    Without a certificate (the drop-in MaskedAdam on a user's own graph) no bitmap exists and the scanning kernels see every
    element: the recycling optimizer equals the non-recycling one and the parked buffer is all zero afterwards.  With a certificate
    wrongly given, a second lookup still voids it."""
    from unboundednerfpytorch_amd import _gradpool
    from unboundednerfpytorch_amd.grid import GridQuery
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    _lib, grid0, pts, go, lo, hi = _setup(seed=5)
    _gradpool.clear()
    try:
        res = []
        for recycle in (True, False):
            p = torch.nn.Parameter(grid0.clone(memory_format=torch.preserve_format))
            opt = MaskedAdam([{"params": [p], "lr": 0.1, "skip_zero_grad": True}], recycle_grads=recycle)
            for it in range(2):
                out = GridQuery.apply(p, pts, lo, hi, 1)
                ((out * go).sum() + 1e-3 * p.pow(2).sum()).backward()        # a regulariser on the raw grid: dense gradient
                assert _gradpool.touch_of(p, p.grad) is None
                assert int((p.grad != 0).sum()) > p.numel() // 2
                opt.step()
                if recycle:
                    assert p.grad is None
                    buf = _gradpool.take(id(p), p.shape, p.stride(), p.device)
                    assert buf is not None and int((buf != 0).sum()) == 0     # the all-zero invariant of the pool survived
                    _gradpool.give(p, buf)
                else:
                    opt.zero_grad(set_to_none=True)
            res.append(p.detach().clone())
        diff = (res[0] - res[1]).abs()
        assert int((diff > 1e-4).sum()) <= max(2, int(1e-5 * diff.numel())), float(diff.max())
        # two lookups of one certified parameter in one graph: the second marking backward voids the certificate
        p = torch.nn.Parameter(grid0.clone(memory_format=torch.preserve_format))
        _gradpool.certify([p])
        (GridQuery.apply(p, pts, lo, hi, 1) * go).sum().backward()
        assert _gradpool.touch_of(p, p.grad) is not None
        (GridQuery.apply(p, pts, lo, hi, 1) * go).sum().backward()
        assert _gradpool.touch_of(p, p.grad) is None
    finally:
        _gradpool.clear()
