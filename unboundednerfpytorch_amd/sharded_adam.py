"""Data-parallel MaskedAdam with the per-voxel optimizer state sharded across the ranks (SURVEY.md section 8e,
training: "rays -- and per-voxel Adam state -- shard embarrassingly across the GPUs of one node").

The reference trains on one GPU (run_train.py:185-296).  With N ranks each rendering N_rand/N of the ray batch,
every rank produces a full-size, mostly-zero gradient for the voxel grids.  Instead of all-reducing 2.7 GB and
running Adam N times on replicated state, each parameter is cut into N contiguous flat ranges:

    reduce_scatter(grad)  ->  rank r holds the summed gradient of range r          (RCCL, (N-1)/N x bytes out)
    MaskedAdam kernels on range r only (exp_avg / exp_avg_sq exist only for it: 1/N of the state memory)
    all_gather(param range) -> every rank has the updated parameter again          ((N-1)/N x bytes out)

A training batch touches a few per cent of a grid's voxels, so the two collectives do not move the dense arrays (3.46 GB each
way for the S3 feature grid: >= 2.2 ms per collective over all seven xGMI links, against a 3 ms single-GPU step).  With
`sparse_exchange` (default) the gradient is exchanged by 256-byte LINE (64 floats of the flat storage order): the ranks
all-gather their touched-line bitmaps (one bit per line, 1.7 MB for that grid), OR them, pack the marked lines of each owner's
range, reduce-scatter the packed rows, update their own range, and -- when the update is a masked one, which changes marked lines
only -- all-gather the updated rows instead of the whole range (`last_exchange` records the bytes of the step).  Same sums, same
update kernels: the results equal the dense collectives' bit for bit.

Small parameters (the rgbnet, 88 KB) are all-reduced and updated redundantly.  The update kernels are the same
three as MaskedAdam's (adam_upd / masked_adam_upd / adam_upd_with_perlr); exact zeros stay exact through the sum,
so the skip_zero_grad semantics are unchanged, and with `average=True` (default, DDP convention: every rank's loss
is a mean over its own rays) the result equals the single-process optimizer stepping on the rank-averaged
gradient bit for bit when N is a power of two.
"""
import weakref

import torch
import torch.distributed as dist

from . import _gradpool
from ._lib import wait_pending as _lib_wait


def _low_priority_stream():
    """the side stream of step(overlap=...): the lowest priority the device offers (torch maps an out-of-range value to
    the nearest valid one), so that the caller's stream -- the next forward's latency-bound kernels -- is served first
    and the bandwidth-bound update fills what is left.

    WHICH of torch's pool streams it gets matters on this hardware: the runtime spreads streams over a few hardware queues, and only
    on some pairs of queues is the low priority HONOURED (the caller's kernels dispatched first, the update filling what is left); on
    the others the two streams simply share the chip, and the bandwidth-bound update slows the caller's chain of short kernels -- the
    step's critical path: the S3 masked step reads 3.2 instead of 2.25 ms, the dense one 7.6 instead of 5.8, depending on nothing but
    how many streams the process had created before (profiles/r06/side_stream_queues.txt).  So the stream is CHOSEN: up to eight
    candidates, each tried once the way the loop uses it -- the caller's stream kept busy with ~1.2 ms of back-to-back kernels, one
    ~0.4 ms streaming kernel on the candidate beside them; the first candidate beside which the caller's queue drains FIRST (its
    kernels were not held up: 1.1-1.2 ms against 1.5 ms shared) is kept (a few ms, once per optimizer).  UGRID_SIDE_STREAM_PICK=0
    takes the first candidate."""
    import os
    first = torch.cuda.Stream(priority=1)
    if os.environ.get("UGRID_SIDE_STREAM_PICK", "1") == "0" or torch.cuda.is_current_stream_capturing():
        return first
    main = torch.cuda.current_stream()
    dev = main.device
    try:
        big = torch.empty(256 << 20, device=dev)          # 1 GiB: one launch of ~10^6 workgroups, ~0.4 ms
        mid = torch.empty(8 << 20, device=dev)            # 32 MB: the caller's back-to-back kernels, ~15 us each
    except RuntimeError:
        return first
    cand, verdicts = first, []
    for trial in range(8):
        if trial:
            cand = torch.cuda.Stream(priority=1)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        cand.wait_stream(main)
        torch.cuda.synchronize(dev)
        ev[0].record(main)                                 # (the clock both sides are read against)
        for _ in range(120):
            mid.mul_(1.0)                                  # the caller's queue: issued faster than it executes
        with torch.cuda.stream(cand):
            big.mul_(1.0)                                  # the candidate's one long kernel, issued while ~3/4 of that is still queued
            ev[1].record(cand)
        ev[2].record(main)
        torch.cuda.synchronize(dev)
        t_side, t_main = ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[2])
        verdicts.append((round(t_side, 3), round(t_main, 3)))
        if t_side >= t_main:                               # the caller's queue drained first: the priority is honoured on this pair
            break
    else:
        cand = first
    big.record_stream(cand)
    _low_priority_stream.last = verdicts                   # (diagnostics: tools/bench_train_step.py prints it)
    return cand


class ShardedMaskedAdam(torch.optim.Optimizer):
    """Same constructor and param-group keys as MaskedAdam (`skip_zero_grad` per group, masked_adam.py:21-41) plus
    the process group.  `ops`: module providing the three update kernels (default: the HIP drop-in
    adam_upd_cuda)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, group=None, average=True,
                 min_shard_numel=1 << 16, ops=None, local_only=False, recycle_grads=False, sparse_exchange=True):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("Invalid beta parameters: {}".format(betas))
        # recycle_grads (OPT-IN; train_utils.create_optimizer_or_freeze_model turns it on for this package's training
        # loop): after step() the gradient buffer of a 5-D grid parameter has been re-zeroed by the update kernel, `.grad`
        # is None and the buffer is parked in _gradpool for the parameter's next backward (no 3.5 GB zero fill per step).
        # Off (the reference's behaviour, masked_adam.py:43-75): `.grad` survives step() untouched.  HIP ops only.
        self.recycle_grads = bool(recycle_grads) and ops is None
        if ops is None:
            from . import adam_upd_cuda as ops
        self.ops = ops
        self.group = group
        self.average = bool(average)
        self.min_shard_numel = int(min_shard_numel)
        self.local_only = bool(local_only)     # MaskedAdam: the reference's single-process optimizer, no collectives
        # sharded parameters: exchange only the 256-byte lines some rank touched (see the module docstring); shards are then
        # whole numbers of lines (64 floats) instead of 4-voxel vectors
        self.sparse_exchange = bool(sparse_exchange)
        self.last_exchange = {}                # id(param) -> bytes this rank put on / took off the wire in the last step()
        self.per_lr = None
        self._alt = {}      # second parameter buffers of the fused TV + Adam pass (not optimizer state: never checkpointed)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    # -- topology ------------------------------------------------------------------------------------
    def _world(self):
        if not self.local_only and dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    @staticmethod
    def _flat(t):
        """1-D view of a dense tensor in STORAGE order (canonical or channel-last grids alike)"""
        if t.is_contiguous():
            return t.view(-1)
        if t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
            return t.permute(0, 2, 3, 4, 1).reshape(-1)
        raise RuntimeError("ShardedMaskedAdam needs dense parameters")

    @staticmethod
    def shard_len(numel, world, align=4):
        """Flat elements per rank: ceil(numel / world) rounded up to the update kernels' 4-voxel vectors."""
        per = -(-numel // world)
        return -(-per // align) * align

    LINE = 64      # floats per exchanged line (256 bytes: the touched-line bitmap's granule, include/ugrid_hip.h)
    MULTI_MAX_NUMEL = 1 << 18      # single process: tensors up to this size (the rgbnet's) share ONE update launch (adam_upd_multi)

    def _shard_len(self, numel, world):
        """flat elements per rank for THIS optimizer: ceil(numel / world) rounded up to the update kernels' 4-voxel vectors.  The sparse
        (touched-line) exchange applies when that split happens to be exact and a whole number of 256-byte lines per rank -- every
        large grid at world 2 / 4 / 8 (_sparse_plan checks it) -- else the dense collectives run on a zero-padded copy"""
        return self.shard_len(numel, world, 4)

    def set_pervoxel_lr(self, count):
        assert self.param_groups[0]['params'][0].shape == count.shape
        self.per_lr = count.float() / count.max()

    # -- one parameter ---------------------------------------------------------------------------------
    def _update(self, group, p, g, m, v, step, per_lr, recycle=None, touch=None):
        """recycle: the grid parameter whose .grad `g` is -- after a masked update on the HIP ops the gradient buffer comes
        back all zero (rezero_grad) and is parked for the parameter's next backward (_gradpool).  touch: the buffer's
        touched-line bitmap (_gradpool.touch_of), used by the rezero path only"""
        beta1, beta2 = group['betas']
        args = (step, beta1, beta2, group['lr'], group['eps'])
        if per_lr is not None:
            self.ops.adam_upd_with_perlr(p, g, m, v, per_lr, *args)
        elif group['skip_zero_grad']:
            rz = getattr(self.ops, 'masked_adam_upd_rezero', None)
            if (rz is not None and recycle is not None and self.recycle_grads and _gradpool.enabled and g is recycle.grad
                    and recycle.dim() == 5 and g.is_cuda and g.stride() == recycle.stride()):
                if touch is not None:
                    rz(p, g, m, v, *args, touch=touch)
                else:
                    rz(p, g, m, v, *args)
                if _gradpool.give(recycle, g):      # parked: the pool now owns the (all-zero) buffer
                    recycle.grad = None
            else:
                self.ops.masked_adam_upd(p, g, m, v, *args)
        else:
            self.ops.adam_upd(p, g, m, v, *args)

    def _touch_of(self, param, g, grad_hook):
        """the touched-line bitmap of `g` when this step may rely on it: single process, HIP ops, recycled gradients, nobody
        edited the gradient (grad_hook) -- see _gradpool"""
        if grad_hook is not None or not self.recycle_grads or self._world()[0] != 1:
            return None
        return _gradpool.touch_of(param, g)

    def _tv_then_update(self, group, param, g, state, tv, use_perlr, side=None, touch=None):
        """Total-variation term (w, dense_mode, tv_module) on the reduced gradient `g`, then the Adam update.  Dense mode
        on the HIP ops: ONE fused pass (adam_upd_cuda.tv_adam_dense -- 7 instead of 13 array transfers, the gradient is
        not written back, bit-identical results); the new parameter values land in a second buffer that is swapped in."""
        w, dense, tv_module = tv
        fused_fn = getattr(self.ops, 'tv_adam_dense', None)
        if dense and fused_fn is not None and tv_module is None and not use_perlr and param.dim() >= 3 and g.stride() == param.stride():
            alt = self._alt.get(param)
            if alt is None or alt.shape != param.shape or alt.device != param.device or alt.stride() != param.stride():
                alt = torch.empty_like(param.data, memory_format=torch.preserve_format)
            beta1, beta2 = group['betas']
            # the gradient buffer comes back all zero and is parked for the next backward (_gradpool): no zero fill per step
            recycle = self.recycle_grads and _gradpool.enabled and g is param.grad
            kw = {'rezero_grad': True} if recycle else {}
            if recycle and touch is not None:
                kw['touch'] = touch
            args = (param.data, alt, g, state['exp_avg'], state['exp_avg_sq'], w, w, w, state['step'], beta1, beta2,
                    group['lr'], group['eps'], group['skip_zero_grad'])
            if side is not None:
                # the 7-pass update of this grid on a second HIP stream: the caller's stream goes on (the next forward's
                # density march, its host syncs, the launch-bound glue) and meets it again at _lib.wait_pending
                side.wait_stream(torch.cuda.current_stream(param.device))
                # every array the side-stream kernel touches was allocated on the caller's stream: tell the caching
                # allocator, so that a buffer whose last reference dies before the pass has run (zero_grad(set_to_none),
                # a pool that refuses it, a replaced `alt`) is not handed to another tensor while it is still in use
                for t_ in (g, alt, param.data, state['exp_avg'], state['exp_avg_sq']) + ((touch,) if 'touch' in kw else ()):
                    t_.record_stream(side)
                with torch.cuda.stream(side):
                    done = fused_fn(*args, **kw)
                    if done:
                        ev = torch.cuda.Event()
                        ev.record(side)
                        param._ug_pending = ev
            else:
                done = fused_fn(*args, **kw)
            if done:
                self._alt[param] = param.data
                param.data = alt
                if recycle and _gradpool.give(param, g):     # parked (all zero once the pass has run); else .grad keeps it alive
                    param.grad = None
                return
        ours = tv_module is None
        if ours:
            from . import total_variation_cuda as tv_module
        if dense:
            touch = None      # the dense term writes every element: the bitmap does not describe the gradient any more

        def run():
            if touch is not None and ours and not dense and g.dim() == 5 and g.shape[1] % 4 == 0:
                tv_module.total_variation_add_grad_touched(param, g, w, w, w, touch)
            else:
                tv_module.total_variation_add_grad(param, g, w, w, w, dense)
            self._update(group, param, g, state['exp_avg'], state['exp_avg_sq'], state['step'],
                         self.per_lr if use_perlr else None, recycle=param, touch=touch)
        if side is not None and ours and touch is not None and not dense:
            # masked TV + masked Adam on the marked lines, queued on the second stream like the dense pass (they update the
            # parameter in place: consumers meet them at _lib.wait_pending)
            side.wait_stream(torch.cuda.current_stream(param.device))
            for t_ in (g, param.data, state['exp_avg'], state['exp_avg_sq'], touch):
                t_.record_stream(side)
            with torch.cuda.stream(side):
                run()
                ev = torch.cuda.Event()
                ev.record(side)
                param._ug_pending = ev
        else:
            run()

    @torch.no_grad()
    def step_param(self, param, tv_term=None, overlap=False):
        """The update of ONE replicated parameter, as step() would do it, callable as soon as its gradient is complete -- e.g.
        from a post-accumulate-grad hook during loss.backward(), so that the largest grid's update (optionally on the side
        stream, overlap=True) starts while the rest of the backward still runs.  The following step() skips the parameter.
        Single process only; returns False (nothing done) otherwise or when the parameter has no gradient."""
        if self._world()[0] != 1 or param.grad is None:
            return False
        group = next((g for g in self.param_groups if any(param is q for q in g['params'])), None)
        if group is None:
            return False
        _lib_wait(param)
        state = self.state[param]
        g = param.grad
        if len(state) == 0:
            state['step'] = 0
            state['exp_avg'] = torch.zeros_like(param, memory_format=torch.preserve_format)
            state['exp_avg_sq'] = torch.zeros_like(param, memory_format=torch.preserve_format)
        state['step'] += 1
        use_perlr = self.per_lr is not None and param.shape == self.per_lr.shape
        touch = self._touch_of(param, g, None)
        if tv_term is not None:
            side = None
            if overlap:
                side = self._side = getattr(self, '_side', None) or _low_priority_stream()
            self._tv_then_update(group, param, g, state, tv_term, use_perlr, side, touch=touch)
        else:
            self._update(group, param, g, state['exp_avg'], state['exp_avg_sq'], state['step'],
                         self.per_lr if use_perlr else None, recycle=param, touch=touch)
        # remembered by OBJECT (a weak reference beside the id): a parameter that is replaced -- scale_volume_grid makes new
        # ones -- may get the id of a dead one, and must not be skipped by the next step() on the strength of that
        self._early = getattr(self, '_early', None) or {}
        self._early[id(param)] = weakref.ref(param)
        return True

    @torch.no_grad()
    def step(self, grad_hook=None, tv_terms=None, overlap=None):
        """grad_hook(param, grad): optional in-place edit of the REDUCED gradient before the update.  For a sharded
        parameter the hook receives a full-shape gradient that is zero outside this rank's range.
        tv_terms: optional {param: (w, dense_mode, tv_module or None)} -- the total-variation term of the training
        iteration (run_train.py:281-287), applied to the gradient summed over all ranks (a rank-local masked TV would
        give a voxel touched by k of N ranks only k/N of the term); for a replicated parameter in dense mode it is
        fused with the update (see _tv_then_update).
        overlap: optional collection of grid parameters whose fused dense TV + Adam pass may run on a second HIP stream
        (single process only).  The caller's stream is not blocked by it; the parameter, its moments and its recycled
        gradient buffer are complete once `_lib.wait_pending(param)` has been called on the consuming stream -- the grid
        modules do that in forward / get_dense_grid / state_dict / scale_volume_grid, this optimizer in step and
        state_dict.  Code that reads `param.data` directly must call it (or torch.cuda.synchronize()) first."""
        world, rank = self._world()
        tv_terms = tv_terms or {}
        self.last_exchange = {}          # (per step: parameters are replaced at every pg_scale event)
        scale = (1.0 / world) if (self.average and world > 1) else None
        side = None
        if overlap and world == 1:
            side = self._side = getattr(self, '_side', None) or _low_priority_stream()
        early, self._early = (getattr(self, '_early', None) or {}), {}
        multi = getattr(self.ops, 'adam_upd_multi', None)
        for group in self.param_groups:
            batch = []        # small tensors of this group, updated by ONE launch at the end of the group (HIP ops only)
            for param in group['params']:
                ref = early.get(id(param))
                if param.grad is None or (ref is not None and ref() is param):   # (updated already by step_param during the backward)
                    continue
                _lib_wait(param)
                state = self.state[param]
                n = param.numel()
                use_perlr = self.per_lr is not None and param.shape == self.per_lr.shape
                sharded = world > 1 and n >= self.min_shard_numel
                if (multi is not None and world == 1 and n <= self.MULTI_MAX_NUMEL and param not in tv_terms and not use_perlr
                        and grad_hook is None and param.is_cuda and param.is_contiguous() and param.grad.is_contiguous()
                        and param.dtype == torch.float32 and param.grad.dtype == torch.float32 and param.dim() != 5
                        and param.grad.device == param.device):
                    if len(state) == 0:
                        state['step'] = 0
                        state['exp_avg'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                        state['exp_avg_sq'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                    state['step'] += 1
                    batch.append((param.data, param.grad, state['exp_avg'], state['exp_avg_sq'], state['step'], group['lr']))
                    continue
                if not sharded:
                    # replicated: every rank applies the same (summed) gradient to its own full state
                    g = param.grad
                    if world > 1:
                        dist.all_reduce(g, group=self.group)
                        if scale is not None:
                            g.mul_(scale)
                    if grad_hook is not None:
                        grad_hook(param, g)
                    if len(state) == 0:
                        state['step'] = 0
                        state['exp_avg'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                        state['exp_avg_sq'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                    state['step'] += 1
                    touch = self._touch_of(param, g, grad_hook)
                    if param in tv_terms:
                        self._tv_then_update(group, param, g, state, tv_terms[param], use_perlr,
                                             side if (side is not None and any(param is q for q in overlap)) else None, touch=touch)
                    else:
                        self._update(group, param, g, state['exp_avg'], state['exp_avg_sq'], state['step'],
                                     self.per_lr if use_perlr else None, recycle=param, touch=touch)
                    continue
                per = self._shard_len(n, world)
                total = per * world
                b = min(n, rank * per)
                e = min(n, b + per)
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros(per, dtype=param.dtype, device=param.device)      # this rank's
                    state['exp_avg_sq'] = torch.zeros(per, dtype=param.dtype, device=param.device)   # range only
                    state['shard'] = (b, e, per)
                state['step'] += 1
                flat_p = self._flat(param.data)
                g_full = param.grad
                if g_full.stride() != param.stride():                    # bring a foreign-layout gradient to the parameter's
                    g_full = torch.empty_like(param.data).copy_(g_full)
                flat_g = self._flat(g_full)
                exact = (total == n)       # no padding: collectives run on the parameter / gradient storage itself
                if not exact:
                    pad_g = torch.zeros(total, dtype=flat_g.dtype, device=flat_g.device)
                    pad_g[:n] = flat_g
                    flat_g = pad_g
                # masked update: only elements with a non-zero (reduced, TV-augmented) gradient move, all inside marked lines
                tv_dense = param in tv_terms and bool(tv_terms[param][1])
                masked = bool(group['skip_zero_grad']) and not use_perlr and not tv_dense and grad_hook is None
                plan = self._sparse_plan(param, flat_g, n, per, world, rank) if (self.sparse_exchange and exact) else None
                if plan is not None:
                    g_shard = self._sparse_reduce_scatter(plan, flat_g, per, world, rank)
                    self.last_exchange[id(param)] = {
                        "mode": "sparse reduce-scatter, dense all-gather", "lines_total": n // self.LINE, "lines_union": int(plan["lines"].numel()),
                        "line_source": getattr(self, "_line_source", None),
                        "rows_per_rank_padded": plan["M"], "bitmap_bytes": plan["bitmap_bytes"],
                        "reduce_scatter_bytes": plan["M"] * world * self.LINE * 4, "all_gather_bytes": per * world * 4,
                        "dense_bytes_each_way": n * 4}
                else:
                    g_shard = torch.empty(per, dtype=flat_g.dtype, device=flat_g.device)
                    dist.reduce_scatter_tensor(g_shard, flat_g, group=self.group)
                    self.last_exchange[id(param)] = {"mode": "dense", "reduce_scatter_bytes": flat_g.numel() * 4,
                                                     "all_gather_bytes": per * world * 4}
                if scale is not None:
                    g_shard.mul_(scale)
                if grad_hook is not None or param in tv_terms:
                    full = torch.zeros_like(param.data)                 # the parameter's own layout
                    full_g = self._flat(full)
                    full_g[b:e] = g_shard[: e - b]
                    if grad_hook is not None:
                        grad_hook(param, full)
                    if param in tv_terms:
                        w, dense, tv_module = tv_terms[param]
                        if tv_module is None:
                            from . import total_variation_cuda as tv_module
                        tv_module.total_variation_add_grad(param, full, w, w, w, dense)
                    g_shard[: e - b] = full_g[b:e]
                    del full_g, full
                if exact:
                    p_shard = flat_p[b:b + per]                  # a view: updated in place
                else:
                    p_shard = torch.zeros(per, dtype=flat_p.dtype, device=flat_p.device)
                    p_shard[: e - b] = flat_p[b:e]
                lr_shard = None
                if use_perlr:
                    lr_shard = torch.zeros(per, dtype=torch.float32, device=param.device)
                    lr_shard[: e - b] = self.per_lr.reshape(-1)[b:e]
                self._update(group, p_shard, g_shard, state['exp_avg'], state['exp_avg_sq'], state['step'], lr_shard)
                if plan is not None and masked:
                    self._sparse_all_gather(plan, flat_p, world, rank)
                    self.last_exchange[id(param)]["all_gather_bytes"] = plan["M"] * world * self.LINE * 4
                    self.last_exchange[id(param)]["mode"] = "sparse"
                elif exact:
                    dist.all_gather_into_tensor(flat_p, p_shard.clone(), group=self.group)
                else:
                    full = torch.empty(total, dtype=flat_p.dtype, device=flat_p.device)
                    dist.all_gather_into_tensor(full, p_shard, group=self.group)
                    flat_p.copy_(full[:n])
            # one launch per DEVICE: the multi-tensor kernel runs on its first item's device and stream (a group that spans
            # devices must not hand it foreign pointers -- ADVICE r5)
            by_dev = {}
            for item in batch:
                by_dev.setdefault(item[0].device, []).append(item)
            for items in by_dev.values():
                multi(items, group['betas'][0], group['betas'][1], group['eps'], bool(group['skip_zero_grad']))

    # -- sparse (touched-line) exchange ----------------------------------------------------------------
    def _line_bits(self, param, flat_g, n):
        """bool [n / LINE]: lines of THIS rank's gradient that hold a non-zero.  From the backward's touched-line bitmap when
        the step certified one (no scan of the 3.46 GB array: stale set bits only cost bytes), else one pass over the gradient."""
        nl = n // self.LINE
        t = _gradpool.touch_of(param, param.grad) if (self.recycle_grads and flat_g.is_cuda) else None
        if t is not None and t.numel() * 32 >= nl:
            w = t.view(-1, 1)
            bits = ((w >> torch.arange(32, device=t.device, dtype=torch.int32)[None, :]) & 1).reshape(-1)[:nl].bool()
            # this gradient buffer is not recycled by the sharded step: its successor starts from a fresh bitmap, and the one just
            # read is DROPPED rather than zeroed in place -- param.grad still holds data, and a zeroed bitmap served to a later
            # touch_of consumer of the same step would claim "every line is zero" 
            _gradpool.consume_touch(param)
            self._line_source = "backward's touched-line bitmap"
            return bits
        self._line_source = "scan of the gradient"
        return flat_g.view(nl, self.LINE).ne(0).any(dim=1)

    def _sparse_plan(self, param, flat_g, n, per, world, rank):
        """Union of the ranks' touched lines and where each goes in the packed buffers; None = use the dense collectives
        (ragged sizes, or so many lines that packing would not pay).  One small all-gather and ONE host read (the row count)."""
        L = self.LINE
        if n % L or per % L or per * world != n:
            return None
        nl = n // L
        mine = self._line_bits(param, flat_g, n)
        # bit-packed for the wire: 1 bit per line (torch has no bitwise-OR all-reduce on NCCL: gather + OR locally)
        pad = (-nl) % 8
        bits8 = torch.cat([mine, mine.new_zeros(pad)]) if pad else mine
        packed = (bits8.view(-1, 8).to(torch.uint8) << torch.arange(8, device=mine.device, dtype=torch.uint8)[None, :]).sum(dim=1, dtype=torch.uint8)
        allb = torch.empty(world * packed.numel(), dtype=torch.uint8, device=packed.device)
        dist.all_gather_into_tensor(allb, packed, group=self.group)
        union = allb.view(world, -1)[0].clone()
        for r in range(1, world):
            union |= allb.view(world, -1)[r]
        ub = ((union.view(-1, 1) >> torch.arange(8, device=union.device, dtype=torch.uint8)[None, :]) & 1).reshape(-1)[:nl].bool()
        lines = torch.nonzero(ub).flatten()                       # sorted global line ids, identical on every rank
        lpr = per // L                                            # lines per rank
        owner = torch.div(lines, lpr, rounding_mode='floor')
        counts = torch.bincount(owner, minlength=world)
        M = int(counts.max()) if lines.numel() else 0             # host read: sizes the packed buffers
        if M == 0 or M * world * 2 > nl:                          # nothing to send, or no saving over the dense collectives
            return None if M else {"M": 0, "lines": lines, "slot": lines, "mine": lines, "lpr": lpr, "bitmap_bytes": int(allb.numel())}
        start = torch.cumsum(counts, 0) - counts
        slot = owner * M + (torch.arange(lines.numel(), device=lines.device) - start[owner])
        sel = owner == rank
        return {"M": M, "lines": lines, "slot": slot, "mine": lines[sel] - rank * lpr, "mine_slot": slot[sel] - rank * M, "lpr": lpr,
                "bitmap_bytes": int(allb.numel())}

    def _sparse_reduce_scatter(self, plan, flat_g, per, world, rank):
        """summed gradient of this rank's range as a dense [per] tensor (zero outside the marked lines)"""
        L, M = self.LINE, plan["M"]
        g_shard = torch.zeros(per, dtype=flat_g.dtype, device=flat_g.device)
        if M:
            send = torch.zeros(world * M, L, dtype=flat_g.dtype, device=flat_g.device)
            send[plan["slot"]] = flat_g.view(-1, L)[plan["lines"]]
            recv = torch.empty(M, L, dtype=flat_g.dtype, device=flat_g.device)
            dist.reduce_scatter_tensor(recv, send, group=self.group)
            g_shard.view(-1, L)[plan["mine"]] = recv[plan["mine_slot"]]
        return g_shard

    def _sparse_all_gather(self, plan, flat_p, world, rank):
        """the updated rows of every rank's marked lines, written into the full parameter"""
        L, M = self.LINE, plan["M"]
        if not M:
            return
        rows = flat_p.view(-1, L)
        send = torch.zeros(M, L, dtype=flat_p.dtype, device=flat_p.device)
        send[plan["mine_slot"]] = rows[plan["mine"] + rank * plan["lpr"]]
        allp = torch.empty(world * M, L, dtype=flat_p.dtype, device=flat_p.device)
        dist.all_gather_into_tensor(allp, send, group=self.group)
        rows[plan["lines"]] = allp[plan["slot"]]

    # -- checkpointing ---------------------------------------------------------------------------------
    def _would_shard(self, param):
        world, _ = self._world()
        return world > 1 and param.numel() >= self.min_shard_numel

    @torch.no_grad()
    def state_dict(self):
        for group in self.param_groups:
            for param in group['params']:
                _lib_wait(param)
        return self._state_dict()

    def _state_dict(self):
        """The reference optimizer's layout (utils.py:70-74 saves optimizer.state_dict() verbatim): `exp_avg` /
        `exp_avg_sq` in the PARAMETER's shape, no shard bookkeeping -- so a checkpoint written by a sharded run loads
        into the reference's MaskedAdam (and into MaskedAdam here) and vice versa.  COLLECTIVE when parameters are
        sharded (an all-gather per state tensor): every rank must call it, every rank gets the full dictionary."""
        sd = super().state_dict()
        index = 0
        for group in self.param_groups:
            for param in group['params']:
                st = self.state.get(param)
                if st is not None and 'shard' in st:
                    exp_avg, exp_avg_sq = self.gather_full_state(param)
                    sd['state'][index] = {'step': st['step'], 'exp_avg': exp_avg, 'exp_avg_sq': exp_avg_sq}
                index += 1
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Accepts the full-shape layout (the reference's, MaskedAdam's, or state_dict() above) under any world size:
        after torch has cast and placed the tensors, a parameter this run shards keeps only its own flat range
        [b, e) (zero-padded to the shard length), exactly what the lazy initialisation of step() would have set up."""
        for st in state_dict.get('state', {}).values():
            if 'shard' in st:
                raise RuntimeError("rank-local optimizer state (written before state_dict() gathered shards) cannot be "
                                   "loaded: it holds one rank's range only")
        super().load_state_dict(state_dict)
        # moments arrive in the file's (canonical) layout; the kernels need them in the parameter's own
        for group in self.param_groups:
            for param in group['params']:
                st = self.state.get(param)
                if not st:
                    continue
                for k in ('exp_avg', 'exp_avg_sq'):
                    if k in st and st[k].shape == param.shape and st[k].stride() != param.stride():
                        st[k] = torch.empty_like(param.data).copy_(st[k])
        world, rank = self._world()
        for group in self.param_groups:
            for param in group['params']:
                st = self.state.get(param)
                if not st or not self._would_shard(param):
                    continue
                n = param.numel()
                per = self._shard_len(n, world)
                b = min(n, rank * per)
                e = min(n, b + per)
                for k in ('exp_avg', 'exp_avg_sq'):
                    full = self._flat(st[k])
                    if full.numel() != n:
                        raise RuntimeError("optimizer state %s has %d elements, parameter has %d" % (k, full.numel(), n))
                    shard = torch.zeros(per, dtype=full.dtype, device=full.device)
                    shard[: e - b] = full[b:e]
                    st[k] = shard
                st['shard'] = (b, e, per)

    @torch.no_grad()
    def gather_full_state(self, param):
        """exp_avg / exp_avg_sq of `param` assembled on every rank in the parameter's shape (what the reference's
        optimizer_state_dict would hold, utils.py:70-74); for saving a checkpoint."""
        world, _ = self._world()
        st = self.state[param]
        if 'shard' not in st:
            return st['exp_avg'], st['exp_avg_sq']
        n = param.numel()
        out = []
        for k in ('exp_avg', 'exp_avg_sq'):
            full = torch.empty(st['shard'][2] * world, dtype=st[k].dtype, device=st[k].device)
            dist.all_gather_into_tensor(full, st[k], group=self.group)
            # the shards follow the parameter's STORAGE order (a channel-last grid is not row-major): put them back through
            # a tensor of the parameter's own strides, so that the result is the right LOGICAL [P,C,X,Y,Z] tensor
            t = torch.empty_like(param.data, memory_format=torch.preserve_format)
            self._flat(t).copy_(full[:n])
            out.append(t)
        return tuple(out)
