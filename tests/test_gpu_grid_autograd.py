"""SURVEY.md section 8 row f2 (first piece): the Fourier / dense grid lookup as an autograd op on the HIP kernels --
forward = ugrid_grid_query, backward = ugrid_grid_query_backward (scatter of the output gradient through the
trilinear weights, hardware fp32 atomics) -- against torch autograd through the oracle's F.grid_sample restatement
of FourierGrid.forward (FourierGrid_grid.py:60-78).  Like torch's own grid_sample backward the device accumulation
order is not fixed, so gradients agree to rounding (tolerance below), not bit for bit."""
import numpy as np
import pytest
import torch

import synth
from oracle import model_oracle

pytestmark = pytest.mark.gpu

GRAD_RTOL = 2e-5   # of the largest |gradient entry| (fp32 sums of up to ~n/cells terms in arbitrary order)


def _case(C, Fq, G, n, seed):
    P = 1 + 2 * Fq if Fq > 0 else 1
    grid = torch.from_numpy(synth.normal(seed, P * C * G[0] * G[1] * G[2]).reshape(P, C, *G))
    pts = torch.from_numpy(synth.uniform(seed + 1, n * 3, -1.4, 1.4).reshape(n, 3))   # some fall outside the box
    pts[:4] = torch.tensor([[-1.2, -1.2, -1.2], [1.2, 1.2, 1.2], [0.0, 0.0, 0.0], [1.3, 0.1, -0.2]])
    gout = torch.from_numpy(synth.normal(seed + 2, n * C).reshape(n, C))
    gout[5:9] = 0.0   # exact zeros must not touch the gradient
    return grid, pts, gout


@pytest.mark.parametrize("C,Fq,G,n", [(1, 3, (9, 7, 5), 4001), (12, 3, (8, 8, 8), 3000), (3, 2, (6, 9, 7), 2000),
                                      (4, 0, (6, 8, 11), 2500), (1, 0, (5, 5, 5), 700)])
def test_grid_query_backward_matches_torch_autograd(C, Fq, G, n):
    from unboundednerfpytorch_amd.grid import GridQuery
    grid, pts, gout = _case(C, Fq, G, n, 900 + C + 10 * Fq)
    lo, hi = torch.full((3,), -1.2), torch.full((3,), 1.2)
    # oracle: torch autograd on CPU
    g_ref = grid.clone().requires_grad_(True)
    out_ref = model_oracle.fourier_grid_query(g_ref, pts, lo, hi, Fq).reshape(n, C)
    (out_ref * gout).sum().backward()
    # product: HIP forward + HIP backward
    g_dev = grid.cuda().requires_grad_(True)
    out = GridQuery.apply(g_dev, pts.cuda(), lo.cuda(), hi.cuda(), Fq).reshape(n, C)
    (out * gout.cuda()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), rtol=0, atol=2e-5)
    ref = g_ref.grad.numpy()
    got = g_dev.grad.cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= GRAD_RTOL * max(1.0, np.abs(ref).max())
    # voxels no sample touches keep an exact zero (MaskedAdam's skip_zero_grad depends on it)
    assert np.array_equal(got == 0, ref == 0) or (np.abs(ref[(got == 0) != (ref == 0)]).max() < 1e-6)


def test_fourier_grid_module_mirrors_the_reference_module():
    from unboundednerfpytorch_amd.grid import FourierGrid, create_grid, grid_query
    ws = torch.tensor([7, 6, 5])
    m = create_grid('DenseGrid', channels=12, world_size=ws, xyz_min=[-1.2] * 3, xyz_max=[1.2] * 3,
                    use_nerf_pos=True, fourier_freq_num=3, config=None).cuda()
    assert isinstance(m, FourierGrid)
    assert set(m.state_dict().keys()) == {"grid", "xyz_min", "xyz_max"}     # FourierGrid_grid.py:43-58
    assert tuple(m.grid.shape) == (7, 12, 7, 6, 5)
    with torch.no_grad():
        m.grid.copy_(torch.from_numpy(synth.normal(5, m.grid.numel()).reshape(m.grid.shape)))
    pts = torch.from_numpy(synth.uniform(6, 300 * 3, -1.2, 1.2).reshape(10, 30, 3)).cuda()
    out = m(pts)
    assert out.shape == (10, 30, 12)
    assert torch.equal(out.detach(), grid_query(m.grid.detach(), pts, m.xyz_min, m.xyz_max, 3))
    out.square().sum().backward()
    assert m.grid.grad is not None and float(m.grid.grad.abs().sum()) > 0
    before = m.grid.grad.clone()
    m.total_variation_add_grad(1e-3, 1e-3, 1e-3, True)                     # in place on .grad, dense mode
    assert not torch.equal(before, m.grid.grad)
    m.scale_volume_grid([9, 8, 7])
    assert tuple(m.grid.shape) == (7, 12, 9, 8, 7)
    m -= 0.5
    d = create_grid('DenseGrid', channels=1, world_size=ws, xyz_min=[-1.0] * 3, xyz_max=[1.0] * 3,
                    use_nerf_pos=False, fourier_freq_num=5, config=None).cuda()
    assert tuple(d.grid.shape) == (1, 1, 7, 6, 5) and d(pts).shape == (10, 30)
    assert "channels=12" in repr(m)


def test_mask_grid_constructor_variants(tmp_path):
    """MaskGrid(path=..., mask_cache_thres=...) / MaskGrid(path=None, mask=..., xyz_min=..., xyz_max=...) as the
    reference models call it (FourierGrid_model.py:259,437; dvgo.py:88-96)."""
    import torch.nn.functional as Fn
    from unboundednerfpytorch_amd.grid import MaskGrid
    from unboundednerfpytorch_amd import render_utils_cuda
    G = (9, 8, 7)
    dens = torch.from_numpy(synth.normal(77, G[0] * G[1] * G[2]).reshape(1, 1, *G)) * 4
    ck = {"model_kwargs": {"xyz_min": [-1.0, -1.0, -1.0], "xyz_max": [1.0, 1.5, 1.0], "voxel_size_ratio": 0.5},
          "model_state_dict": {"density.grid": dens, "act_shift": torch.tensor([-2.0])}}
    path = str(tmp_path / "coarse_last.tar")
    torch.save(ck, path)
    m = MaskGrid(path=path, mask_cache_thres=0.9).cuda()
    pooled = Fn.max_pool3d(dens, kernel_size=3, padding=1, stride=1)
    want = (1 - torch.exp(-Fn.softplus(pooled - 2.0) * 0.5) >= 0.9)[0, 0]
    assert torch.equal(m.mask.cpu(), want) and 0 < int(want.sum()) < want.numel()
    m2 = MaskGrid(path=None, mask=want, xyz_min=[-1.0, -1.0, -1.0], xyz_max=[1.0, 1.5, 1.0]).cuda()
    pts = torch.from_numpy(synth.uniform(78, 500 * 3, -1.2, 1.6).reshape(5, 100, 3)).cuda()
    a, b = m(pts), m2(pts)
    assert a.shape == (5, 100) and a.dtype == torch.bool and torch.equal(a, b)
    direct = render_utils_cuda.maskcache_lookup(m.mask, pts.reshape(-1, 3).contiguous(), m.xyz2ijk_scale, m.xyz2ijk_shift)
    assert torch.equal(a.flatten(), direct)
    assert "mask.shape=[9, 8, 7]" in repr(m)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last_3d)


@pytest.mark.parametrize("shape,F", [((7, 12, 9, 7, 6), 3), ((1, 4, 5, 6, 8), 0), ((5, 8, 6, 6, 6), 2), ((3, 4, 53, 9, 11), 1)])
def test_channel_last_grid_layout_equals_canonical(shape, F):
    """The training layout of multi-channel grids ([P][X][Y][Z][C] = torch.channels_last_3d of the same logical tensor):
    lookup bit-identical to the canonical layout, scatter backward equal up to the atomics' summation order and produced
    in the grid's own layout, TV gradient (dense and masked) and the fused dense TV + Adam pass bit-identical."""
    from unboundednerfpytorch_amd import adam_upd_cuda, total_variation_cuda
    from unboundednerfpytorch_amd.grid import GridQuery, grid_query
    P, C, X, Y, Z = shape
    n = int(np.prod(shape))
    g = torch.from_numpy(synth.normal(5, n).reshape(shape)).cuda()
    pts = torch.from_numpy(synth.uniform(6, 3000 * 3, -1.3, 1.3).reshape(3000, 3)).cuda()
    lo, hi = torch.full((3,), -1.2).cuda(), torch.full((3,), 1.2).cuda()
    a = grid_query(g, pts, lo, hi, F)
    b = grid_query(_cl(g), pts, lo, hi, F)
    assert torch.equal(a, b)
    go = torch.from_numpy(synth.normal(7, 3000 * C).reshape(3000, C)).cuda()
    go[::5] = 0
    ga = g.clone().requires_grad_(True)
    gb = _cl(g).clone(memory_format=torch.preserve_format).requires_grad_(True)
    (GridQuery.apply(ga, pts, lo, hi, F) * go).sum().backward()
    (GridQuery.apply(gb, pts, lo, hi, F) * go).sum().backward()
    assert gb.grad.stride() == gb.stride() and not gb.grad.is_contiguous()
    scale = float(ga.grad.abs().max())
    assert float((ga.grad - gb.grad).abs().max()) <= 2e-6 * scale
    assert torch.equal(ga.grad != 0, gb.grad != 0)
    # TV gradient
    for dense in (True, False):
        gr = torch.from_numpy(synth.normal(8, n).reshape(shape)).cuda()
        gr[gr.abs() < 0.8] = 0
        gr_a, gr_b = gr.clone(), _cl(gr).clone(memory_format=torch.preserve_format)
        total_variation_cuda.total_variation_add_grad(g, gr_a, 0.3, 0.3, 0.3, dense)
        total_variation_cuda.total_variation_add_grad(_cl(g), gr_b, 0.3, 0.3, 0.3, dense)
        assert torch.equal(gr_a, gr_b)
    with pytest.raises(RuntimeError, match="mix the canonical and the channel-last"):
        total_variation_cuda.total_variation_add_grad(_cl(g), gr.clone(), 0.3, 0.3, 0.3, True)
    # fused dense TV + Adam, and the Adam kernels on channel-last storage
    m = torch.from_numpy(synth.normal(9, n, 0, 0.1).reshape(shape)).cuda()
    v = torch.from_numpy(synth.uniform(10, n, 0, 0.01).reshape(shape)).cuda()
    gr = torch.from_numpy(synth.normal(11, n).reshape(shape)).cuda()
    args = (4, 0.9, 0.99, 0.1, 1e-8)
    pa, ma, va = g.clone(), m.clone(), v.clone()
    gra = gr.clone()
    total_variation_cuda.total_variation_add_grad(pa, gra, 0.2, 0.2, 0.2, True)
    adam_upd_cuda.masked_adam_upd(pa, gra, ma, va, *args)
    pb, mb, vb = _cl(g), _cl(m).clone(memory_format=torch.preserve_format), _cl(v).clone(memory_format=torch.preserve_format)
    out = torch.empty_like(pb)
    assert out.stride() == pb.stride()
    assert adam_upd_cuda.tv_adam_dense(pb, out, _cl(gr), mb, vb, 0.2, 0.2, 0.2, *args, True)
    assert torch.equal(out, pa) and torch.equal(mb, ma) and torch.equal(vb, va)
    pc, mc, vc = _cl(g).clone(memory_format=torch.preserve_format), _cl(m).clone(memory_format=torch.preserve_format), _cl(v).clone(memory_format=torch.preserve_format)
    pd, md, vd = g.clone(), m.clone(), v.clone()
    adam_upd_cuda.masked_adam_upd(pc, _cl(gr), mc, vc, *args)
    adam_upd_cuda.masked_adam_upd(pd, gr, md, vd, *args)
    assert torch.equal(pc, pd) and torch.equal(vc, vd)
