#!/usr/bin/env python
"""How far apart are two runs of the SAME training step?  (VERDICT r5 item 2 / ADVICE r5 medium.)

The grid lookups' backward scatters with hardware fp32 atomics: the same terms, summed in an order that differs run to run.  The
two native-step tests (tests/test_gpu_train_scale.py::test_native_step_equals_the_op_by_op_step, tests/test_gpu_voxgo_train.py::
test_native_step_equals_the_op_by_op_step) compare native_step.VoxGOStep with the op-by-op step; their bounds must come from the
spread two op-by-op runs show against EACH OTHER, not from a guess.  For every configuration of those tests this tool repeats, `--reps`
times with fresh models of the same seed:

  * one forward + backward of the native step (A), of the op-by-op step (B) and of the op-by-op step again (B');
    per grid parameter   max|gA - gB| / max|gB|   and the floor   max|gB - gB'| / max|gB|;   whether the forward arrays and the
    fixed-order gradients (rgbnet) are bit-identical;
  * the tests' short training trajectories for A, B, B': the largest relative loss difference over the steps, and per parameter the
    largest |difference| in learning-rate steps and the fraction of entries further apart than 2 % of a step.

Writes one JSON document (default profiles/r06/native_step_spread.json): per configuration the per-repetition numbers, their maxima,
and `bounds` = what the tests assert (<= 4 x the observed native-vs-op maximum, and never below the op-vs-op floor).
"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _grads(m):
    return {k: p.grad.clone() for k, p in m.named_parameters()}


def _grad_metrics(ga, gb):
    """per parameter: grid -> max|d| / max|gb|; other -> bit-equal?"""
    grid, fixed_equal = {}, True
    for k in ga:
        if "grid" in k:
            scale = float(gb[k].abs().max()) + 1e-30
            grid[k] = float((ga[k] - gb[k]).abs().max()) / scale
        else:
            fixed_equal = fixed_equal and bool(torch.equal(ga[k], gb[k]))
    return grid, fixed_equal


def _fwd_equal(oa, ob):
    ok = True
    for k in ob:
        if k in oa and torch.is_tensor(ob[k]):
            ok = ok and bool(torch.equal(oa[k].detach(), ob[k].detach()))
    return ok


def _traj_metrics(ra, rb):
    (la, pa), (lb, pb) = ra, rb
    la, lb = np.array(la, dtype=np.float64), np.array(lb, dtype=np.float64)
    out = {"loss_rel_max": float(np.max(np.abs(la - lb) / np.maximum(np.abs(lb), 1e-30))), "first_loss_equal": bool(la.flat[0] == lb.flat[0]), "params": {}}
    for k in pa:
        diff = (pa[k] - pb[k]).abs()
        lr = 0.1 if "grid" in k else 1e-3
        out["params"][k] = {"max_in_lr_steps": float(diff.max()) / lr, "frac_above_2pct_of_a_step": float((diff > 0.02 * lr).sum()) / diff.numel(),
                            "count_above_2pct_of_a_step": int((diff > 0.02 * lr).sum()), "numel": diff.numel()}
    return out


def fourier_case(rand_bkgd, dev):
    """tests/test_gpu_train_scale.py::test_native_step_equals_the_op_by_op_step"""
    import bench_train_step as bts
    from unboundednerfpytorch_amd import ops, train_step as ts
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model
    torch.manual_seed(0)
    m0 = bts.make_model(100, 4, dev, fused=True)          # = tests/test_gpu_train_scale.py build(): G = 100, F = 4
    cfg = dict(bts.TRUCK_CFG)
    cfg.update(weight_nearclip=0.3, weight_distortion=0.01, weight_rgbper=0.01, weight_entropy_last=0.001)
    o, d, v, rgb = bts.random_rays(3000, dev, seed=8)
    kw = dict(stepsize=0.5, rand_bkgd=rand_bkgd)
    coef = ops.loss_coefficients(cfg, len(o), m0.sample_table(0.5, dev).numel(), 0.2, 1)
    models = []
    for native in (True, False, False):
        m = copy.deepcopy(m0)
        m.native_step = native
        models.append(m)
    outs = []
    for m in models:
        torch.manual_seed(5)
        out = m(o, d, v, global_step=1, is_train=True, fused_loss={"target": rgb, "coef": coef}, **kw)
        out["loss"].backward()
        outs.append((out, _grads(m)))
        m.zero_grad(set_to_none=True)
    res = {"native_is_native": type(outs[0][0]["loss"].grad_fn).__name__.startswith("VoxGOStep"),
           "fwd_equal": _fwd_equal(outs[0][0], outs[1][0]), "samples": int(outs[1][0]["weights"].numel())}
    res["grid_grad_native_vs_op"], res["fixed_grads_equal_native_vs_op"] = _grad_metrics(outs[0][1], outs[1][1])
    res["grid_grad_op_vs_op"], res["fixed_grads_equal_op_vs_op"] = _grad_metrics(outs[2][1], outs[1][1])
    traj = []
    for m in models:
        torch.manual_seed(11)
        opt = create_optimizer_or_freeze_model(m, bts.TRUCK_CFG, global_step=0)
        losses = []
        for s in (1, 2, 3, 4):
            oo, dd, vv, tt = bts.random_rays(2048, dev, seed=30 + s)
            losses.append(ts.train_iteration(m, opt, oo, dd, vv, tt, bts.TRUCK_CFG, s, kw, overlap_k0_update=True))
        sd = m.state_dict()
        traj.append((losses, {k: x.detach().clone() for k, x in sd.items() if x.dtype == torch.float32}))
    res["traj_native_vs_op"] = _traj_metrics(traj[0], traj[1])
    res["traj_op_vs_op"] = _traj_metrics(traj[2], traj[1])
    return res


def voxgo_case(kind, dev):
    """tests/test_gpu_voxgo_train.py::test_native_step_equals_the_op_by_op_step"""
    import synth
    import test_gpu_voxgo_train as T
    from unboundednerfpytorch_amd import train_step as ts
    from unboundednerfpytorch_amd.ops import loss_coefficients
    from unboundednerfpytorch_amd.train_utils import create_optimizer_or_freeze_model
    torch.manual_seed(0)
    case = T.DVGO_CASES[0] if kind == "dvgo" else synth.DCVGO_CASES[0]
    m0, name, (o, d, v), kw, R, seed = T.build(kind, case, dev)
    target = torch.from_numpy(synth.uniform(seed + 5, R * 3).reshape(R, 3)).to(dev) * 0.5 + 0.25
    rk = {k: kw[k] for k in kw if k != "render_depth"}
    if kind == "dcvgo":
        rk["rand_bkgd"] = True
    cfg = dict(lrate_density=1e-1, lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_decay=20, pg_scale=[], weight_main=1.0, weight_entropy_last=0.01,
               weight_rgbper=0.01, weight_nearclip=0.0, weight_distortion=0.01 if kind == "dcvgo" else 0.0, tv_every=1, tv_after=0,
               tv_before=7, tv_dense_before=4, weight_tv_density=1e-5, weight_tv_k0=1e-6, skip_zero_grad_fields=['density', 'k0'])
    coef = loss_coefficients(cfg, R, m0.sample_table(rk["stepsize"], dev).numel(), None, 1)
    models = []
    for native in (True, False, False):
        m = copy.deepcopy(m0)
        m.native_step = native
        models.append(m)
    outs = []
    for m in models:
        torch.manual_seed(5)
        out = m(o, d, v, global_step=1, is_train=True, fused_loss={'target': target, 'coef': coef}, **rk)
        out["loss"].backward()
        outs.append((out, _grads(m)))
        m.zero_grad(set_to_none=True)
    res = {"native_is_native": type(outs[0][0]["loss"].grad_fn).__name__.startswith("VoxGOStep"),
           "fwd_equal": _fwd_equal(outs[0][0], outs[1][0]), "samples": int(outs[1][0]["weights"].numel())}
    res["grid_grad_native_vs_op"], res["fixed_grads_equal_native_vs_op"] = _grad_metrics(outs[0][1], outs[1][1])
    res["grid_grad_op_vs_op"], res["fixed_grads_equal_op_vs_op"] = _grad_metrics(outs[2][1], outs[1][1])
    traj = []
    for m in models:
        torch.manual_seed(11)
        opt = create_optimizer_or_freeze_model(m, cfg, global_step=0)
        losses = [ts.train_iteration(m, opt, o, d, v, target, cfg, step, rk) for step in range(1, 9)]
        torch.cuda.synchronize()
        traj.append((losses, {k: p.detach().clone() for k, p in m.named_parameters()}))
    res["traj_native_vs_op"] = _traj_metrics(traj[0], traj[1])
    res["traj_op_vs_op"] = _traj_metrics(traj[2], traj[1])
    return res


def summarise(reps):
    """maxima over the repetitions + the bounds the tests assert"""
    def mx(path):
        vals = []
        for r in reps:
            x = r
            for k in path:
                x = x[k]
            vals.append(max(x.values()) if isinstance(x, dict) else x)
        return max(vals)

    def pmax(which, field):
        return max(max(p[field] for p in r[which]["params"].values()) for r in reps)

    s = {"reps": len(reps), "all_native": all(r["native_is_native"] for r in reps), "fwd_always_equal": all(r["fwd_equal"] for r in reps),
         "fixed_grads_always_equal_native_vs_op": all(r["fixed_grads_equal_native_vs_op"] for r in reps),
         "fixed_grads_always_equal_op_vs_op": all(r["fixed_grads_equal_op_vs_op"] for r in reps),
         "grid_grad_native_vs_op_max": mx(("grid_grad_native_vs_op",)), "grid_grad_op_vs_op_max": mx(("grid_grad_op_vs_op",)),
         "loss_rel_native_vs_op_max": mx(("traj_native_vs_op", "loss_rel_max")), "loss_rel_op_vs_op_max": mx(("traj_op_vs_op", "loss_rel_max")),
         "first_loss_always_equal": all(r["traj_native_vs_op"]["first_loss_equal"] for r in reps),
         "param_max_lr_steps_native_vs_op": pmax("traj_native_vs_op", "max_in_lr_steps"), "param_max_lr_steps_op_vs_op": pmax("traj_op_vs_op", "max_in_lr_steps"),
         "param_frac_above_2pct_native_vs_op": pmax("traj_native_vs_op", "frac_above_2pct_of_a_step"),
         "param_frac_above_2pct_op_vs_op": pmax("traj_op_vs_op", "frac_above_2pct_of_a_step"),
         "param_count_above_2pct_native_vs_op": pmax("traj_native_vs_op", "count_above_2pct_of_a_step"),
         "param_count_above_2pct_op_vs_op": pmax("traj_op_vs_op", "count_above_2pct_of_a_step")}
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "native_step_spread.json"))
    ap.add_argument("--cases", default="fourier_False,fourier_True,dvgo,dcvgo")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    doc = {"what": __doc__.split("\n\n")[0], "reps": a.reps, "cases": {}}
    for name in a.cases.split(","):
        reps = []
        for i in range(a.reps):
            if name.startswith("fourier"):
                reps.append(fourier_case(name.endswith("True"), dev))
            else:
                reps.append(voxgo_case(name, dev))
            torch.cuda.empty_cache()
        doc["cases"][name] = {"summary": summarise(reps), "per_rep": reps}
        print(name, json.dumps(doc["cases"][name]["summary"]), flush=True)
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(doc, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
