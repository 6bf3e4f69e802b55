"""Data-parallel training on the REAL kernels with two ranks that share one GPU (gloo; debugging set-up for a single-GPU
box -- production is one rank per GPU over RCCL): fourier_model.FourierGridModel (channel-last k0, fused stage 1, RenderLoss)
+ ShardedMaskedAdam (gradient reduce-scatter, shard update in storage order, all-gather, TV on the reduced gradient), three
iterations on half batches with world_size = 2, against the single-process run on the whole batch.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 \\
      tools/dp_train_2rank_shared_gpu.py

Prints one JSON line (rank 0): per parameter the fraction of elements further than 2 % of a learning-rate step from the
single-process result (fp32 atomics and the reduction order differ), plus which collectives gloo ran natively."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--rays", type=int, default=2048, help="global batch (split over the ranks)")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import bench_train_step as bts
    from unboundednerfpytorch_amd import train_step as ts
    from unboundednerfpytorch_amd.sharded_adam import ShardedMaskedAdam
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model
    notes = {}
    # gloo may lack reduce_scatter / all_gather_into_tensor for device tensors: emulate them for THIS debugging run only
    probe = torch.ones(4 * world, device=dev)
    out = torch.empty(4, device=dev)
    try:
        dist.reduce_scatter_tensor(out, probe)
        notes["reduce_scatter_tensor"] = "native"
    except Exception as e:       # noqa: BLE001
        notes["reduce_scatter_tensor"] = "emulated (all_reduce + slice): %s" % str(e)[:80]

        def rs(output, input, group=None, **kw):
            tmp = input.clone()
            dist.all_reduce(tmp, group=group)
            n = output.numel()
            output.copy_(tmp[rank * n:(rank + 1) * n])
        dist.reduce_scatter_tensor = rs
    try:
        dist.all_gather_into_tensor(probe, out)
        notes["all_gather_into_tensor"] = "native"
    except Exception as e:       # noqa: BLE001
        notes["all_gather_into_tensor"] = "emulated (all_gather): %s" % str(e)[:80]

        def ag(output, input, group=None, **kw):
            parts = [torch.empty_like(input) for _ in range(world)]
            dist.all_gather(parts, input.contiguous(), group=group)
            output.copy_(torch.cat(parts))
        dist.all_gather_into_tensor = ag

    cfg = dict(bts.TRUCK_CFG)
    cfg.update(weight_nearclip=0.01, N_rand=a.rays)
    G, F, near = a.grid, 4, 0.2
    NR, per = a.rays, a.rays // world
    rk = dict(stepsize=0.5, rand_bkgd=False)
    torch.manual_seed(0)
    m = bts.make_model(G, F, dev, fused=True)
    opt = create_optimizer_or_freeze_model(m, cfg, 0, sharded=True)
    assert isinstance(opt, ShardedMaskedAdam) and not m.k0.grid.is_contiguous()
    steps = (1, 2, 10001)        # two dense-TV iterations and one in the masked-TV phase
    exchanges = []
    for i, s in enumerate(steps):
        o, d, v, rgb = bts.random_rays(NR, dev, seed=40 + i)
        sl = slice(rank * per, (rank + 1) * per)
        ts.train_iteration(m, opt, o[sl].contiguous(), d[sl].contiguous(), v[sl].contiguous(), rgb[sl].contiguous(), cfg, s, rk,
                           near_thres=near, world_size=world)
        exchanges.append({k: v for k, v in (opt.last_exchange.get(id(m.k0.grid)) or {}).items() if k in ("mode", "line_source", "lines_union", "lines_total")})
    sharded_state = opt.state[m.k0.grid]["exp_avg"].numel() < m.k0.grid.numel()
    sd = opt.state_dict()        # collective: full-shape moments on every rank
    res = None
    if rank == 0:
        torch.manual_seed(0)
        ref = bts.make_model(G, F, dev, fused=True)
        ropt = create_optimizer_or_freeze_model(ref, cfg, 0)
        for i, s in enumerate(steps):
            o, d, v, rgb = bts.random_rays(NR, dev, seed=40 + i)
            ts.train_iteration(ref, ropt, o, d, v, rgb, cfg, s, rk, near_thres=near, world_size=1)
        torch.cuda.synchronize()
        frac, worst = {}, {}
        for (k, a), (_, b) in zip(m.named_parameters(), ref.named_parameters()):
            lr = 0.1 if "grid" in k else 1e-3
            diff = (a - b).abs()
            frac[k] = float((diff > 0.02 * lr).float().mean())
            worst[k] = float(diff.max())
        rsd = ropt.state_dict()
        # relative to each tensor's own scale (the moments of the grids are ~1e-7): a layout mix-up must not hide in an absolute bound
        mom = max(float((sd["state"][i]["exp_avg"].cpu() - rsd["state"][i]["exp_avg"].cpu()).abs().max())
                  / (float(rsd["state"][i]["exp_avg"].abs().max()) + 1e-30) for i in rsd["state"])
        res = {"world": world, "sharded_k0_state": bool(sharded_state), "k0_channels_last": True, "collectives": notes, "k0_exchange_per_step": exchanges,
               "frac_elements_off_by_more_than_2pct_of_a_step": frac, "max_abs_param_diff": worst, "max_rel_exp_avg_diff": mom,
               "ok": bool(max(frac.values()) < 1e-4 and mom < 1e-2)}
    dist.barrier()
    if rank == 0:
        print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
