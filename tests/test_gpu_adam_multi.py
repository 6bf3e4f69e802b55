"""ugrid_adam_upd_multi (round 5): the small tensors of a model updated by ONE launch -- bit-identical to the per-tensor calls the
reference's optimizer makes (masked_adam.py:43-75), and ShardedMaskedAdam / MaskedAdam take it for the rgbnet without changing a
bit of the trajectory."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tensors(seed, shapes, sparse):
    g = torch.Generator(device="cuda").manual_seed(seed)
    ps = [torch.randn(s, device="cuda", generator=g) for s in shapes]
    gs = [torch.randn(s, device="cuda", generator=g) for s in shapes]
    if sparse:
        gs = [x * (torch.rand(x.shape, device="cuda", generator=g) < 0.3) for x in gs]
    return ps, gs


@pytest.mark.parametrize("masked", [False, True])
def test_adam_upd_multi_equals_the_per_tensor_calls_bit_for_bit(masked):
    from unboundednerfpytorch_amd import adam_upd_cuda as ops
    shapes = [(128, 39), (128,), (128, 128), (128,), (3, 128), (3,), (1,), (257, 5), (4096, 17)] + [(7, 3)] * 12      # > 16 tensors: two launches
    ps, gs = _tensors(1, shapes, masked)
    a = [p.clone() for p in ps]
    b = [p.clone() for p in ps]
    ma, va = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    mb, vb = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
    for step in (1, 2, 7):
        lrs = [1e-3 * (1 + i % 3) for i in range(len(ps))]
        for i in range(len(ps)):
            (ops.masked_adam_upd if masked else ops.adam_upd)(a[i], gs[i], ma[i], va[i], step + i % 2, 0.9, 0.99, lrs[i], 1e-8)
        ops.adam_upd_multi([(b[i], gs[i], mb[i], vb[i], step + i % 2, lrs[i]) for i in range(len(ps))], 0.9, 0.99, 1e-8, masked)
        torch.cuda.synchronize()
        for i in range(len(ps)):
            assert torch.equal(a[i], b[i]) and torch.equal(ma[i], mb[i]) and torch.equal(va[i], vb[i]), (step, i)
    assert not torch.equal(a[0], ps[0])


def test_masked_adam_takes_the_multi_tensor_path_for_small_parameters():
    """the optimizer class with and without the one-launch path: same parameters after 5 steps, bit for bit; the 5-D grid keeps its
    own (masked / recycling) update"""
    from unboundednerfpytorch_amd import adam_upd_cuda as ops
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    g = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda: [torch.nn.Parameter(torch.randn(s, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11 + i)))
                  for i, s in enumerate([(1, 4, 8, 8, 8), (128, 39), (128,), (3, 128), (3,)])]
    P1, P2 = mk(), mk()
    grp = lambda P: [{"params": [P[0]], "lr": 0.1, "skip_zero_grad": True}, {"params": P[1:], "lr": 1e-3, "skip_zero_grad": False}]
    o1, o2 = MaskedAdam(grp(P1)), MaskedAdam(grp(P2))
    o2.MULTI_MAX_NUMEL = -1                     # instance override: every parameter through the per-tensor calls
    calls = []
    real = ops.adam_upd_multi
    try:
        ops.adam_upd_multi = lambda items, *a, **k: (calls.append(len(items)), real(items, *a, **k))[1]
        for it in range(5):
            for pa, pb in zip(P1, P2):
                gr = torch.randn(pa.shape, device="cuda", generator=g)
                if pa.dim() == 5:
                    gr = gr * (torch.rand(pa.shape, device="cuda", generator=g) < 0.2)
                pa.grad, pb.grad = gr.clone(), gr.clone()
            o1.step(); o2.step()
    finally:
        ops.adam_upd_multi = real
    torch.cuda.synchronize()
    assert calls == [4] * 5, calls                                   # the four rgbnet-like tensors in one launch per step
    for pa, pb in zip(P1, P2):
        assert torch.equal(pa.data, pb.data)
