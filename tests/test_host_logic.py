"""CPU tests of the host-side logic around the kernels: sample tables, sharding, the 2-rank exchange."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import model_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sample_table_matches_oracle():
    from unboundednerfpytorch_amd.fourier_render import sample_table
    for world_len, stepsize in ((200, 0.5), (200, 1.31), (300, 0.5), (16, 0.7)):
        t, s = sample_table(world_len, stepsize, 0.2)
        t_ref = model_oracle.sample_t(world_len, stepsize, 0.2)
        assert torch.equal(t, t_ref)
        assert torch.equal(s, 1 - 1 / (1 + t_ref))
    assert sample_table(200, 1.31, 0.2)[0].numel() == 256   # S1: N_inner = N_outer = 128
    assert sample_table(200, 0.5, 0.2)[0].numel() == 668    # garden_single.py


def test_shard_bounds_cover_and_align():
    from unboundednerfpytorch_amd.dist import shard_bounds
    for n in (0, 1, 63, 64, 65, 1000, 2073600):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
                assert e0 == b1
            for b, e in spans:
                if e < n:  # every shard that is not the tail is a whole number of 64-ray wave tiles
                    assert (e - b) % 64 == 0
    assert shard_bounds(2073600, 8, 3) == (3 * 259200, 4 * 259200)


def _fake_forward(o, d, v, **kw):
    # stands in for FourierGridRenderer.forward on CPU: any per-ray function works for the exchange logic
    rgb = torch.stack([o[:, 0] + d[:, 0], o[:, 1] * 2, v[:, 2] - 1], dim=1)
    return {"rgb_marched": rgb, "depth": o.sum(-1), "alphainv_last": d.sum(-1)}


def _worker(rank, world, port, R, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from unboundednerfpytorch_amd.dist import render_sharded
    g = torch.Generator().manual_seed(3)
    o, d, v = [torch.randn(R, 3, generator=g) for _ in range(3)]
    out = render_sharded(_fake_forward, o, d, v, stepsize=0.5)
    ref = _fake_forward(o, d, v)
    ok = all(torch.equal(out[k], ref[k]) for k in ref)
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("R", [1000, 64, 7])
def test_render_sharded_two_ranks_gloo(R):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + R) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, R, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
