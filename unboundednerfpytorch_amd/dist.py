"""Multi-GPU render: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Rays are independent and the grids + rgbnet are read-only at render time, so the
data path is: every rank holds a full replica of the bricks, renders its contiguous ray shard with the
fused kernels, and ONE all-gather exchanges the rendered tiles [rgb(3), depth, alphainv_last] = 20 B/ray.
There is no collective inside the march/shade kernels.

`render_sharded(..., interleave=True)` deals 64-ray tiles round-robin instead of in contiguous ranges: on real
scenes sky and dense regions cluster in the image, and the fused kernels' time per tile follows the survivor
count, so contiguous row bands can be badly unbalanced.

`composite_blocks` is the one-model-per-GPU layout of BASELINE.json configs[4] (Waymo-style blocks): every rank
renders ITS block for ALL rays and one all-reduce(sum) of [R+1,5] merges them.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank, align=64):
    """Contiguous shard [b,e) of n rays for `rank`; shard sizes are multiples of `align` (one wave tile)
    except the last.  Deterministic and identical on every rank."""
    per = -(-n // world_size)
    per = -(-per // align) * align
    b = min(n, rank * per)
    e = min(n, b + per)
    return b, e


def tile_assignment(n, world_size, rank, tile=64, group=1):
    """Indices of the rays of `rank` when the list's 64-ray tiles are dealt round-robin in groups of `group` consecutive tiles
    (group k -> rank k % world_size; group = 1: single tiles).  In render_view's / the bench's 8 x 8 pixel-block ray order a tile
    is one pixel block and consecutive tiles are horizontally adjacent blocks, so group = W / 8 deals whole block ROWS: a rank's rays
    then share grid cells with their horizontal neighbours (what the kernels' cache efficiency rests on) while the shares stay
    balanced over the image; group = 1 gives the finest balance and the least sharing (measured: profiles/r05/scaling_proxy.json)."""
    n_tiles = -(-n // tile)
    n_groups = -(-n_tiles // group)
    mine = torch.arange(rank, max(n_groups, rank), world_size)    # (a rank beyond the last group gets none: arange(r, r) is empty)
    tiles = (mine[:, None] * group + torch.arange(group)[None, :]).reshape(-1)
    idx = (tiles[:, None] * tile + torch.arange(tile)[None, :]).reshape(-1)
    return idx[idx < n]


def render_sharded(renderer_forward, rays_o, rays_d, viewdirs, group=None, interleave=False, deal_group=1, **render_kwargs):
    """Render rays [R,3] split across the process group; every rank returns the full
    {'rgb_marched','depth','alphainv_last'}.  `renderer_forward(o,d,v,**kw)` is FourierGridRenderer.forward
    (or any callable with the reference forward's signature and return keys).  interleave: deal tiles round-robin in groups of
    `deal_group` consecutive 64-ray tiles (tile_assignment) instead of contiguous ranges."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rk = dist.get_rank(group) if dist.is_initialized() else 0
    R = rays_o.shape[0]
    kw = dict(render_kwargs)
    kw["render_depth"] = True
    if interleave and ws > 1:
        idx = tile_assignment(R, ws, rk, group=deal_group).to(rays_o.device)
        n_mine = idx.numel()
        per = tile_assignment(R, ws, 0, group=deal_group).numel()      # rank 0 always holds the largest share
        o_, d_, v_ = rays_o[idx].contiguous(), rays_d[idx].contiguous(), viewdirs[idx].contiguous()
    else:
        b, e = shard_bounds(R, ws, rk)
        n_mine = e - b
        per = shard_bounds(R, ws, 0)[1]
        o_, d_, v_ = rays_o[b:e].contiguous(), rays_d[b:e].contiguous(), viewdirs[b:e].contiguous()
    tile = torch.zeros(per, 5, dtype=torch.float32, device=rays_o.device)
    if n_mine > 0:
        out = renderer_forward(o_, d_, v_, **kw)
        tile[:n_mine, 0:3] = out["rgb_marched"]
        tile[:n_mine, 3] = out["depth"]
        tile[:n_mine, 4] = out["alphainv_last"]
    if ws > 1:
        full = torch.empty(ws * per, 5, dtype=torch.float32, device=rays_o.device)
        dist.all_gather_into_tensor(full, tile, group=group)
        if interleave:
            # undo the deal: rank r's rows are its tiles r, r+ws, ... in order
            res = torch.empty(R, 5, dtype=torch.float32, device=rays_o.device)
            for r in range(ws):
                ir = tile_assignment(R, ws, r, group=deal_group).to(rays_o.device)
                res[ir] = full[r * per: r * per + ir.numel()]
            full = res
    else:
        full = tile
    full = full[:R]
    return {"rgb_marched": full[:, 0:3].contiguous(), "depth": full[:, 3].contiguous(),
            "alphainv_last": full[:, 4].contiguous()}


INVISIBLE_SCALE = 2.0 ** -60   # weight factor of a block that fails the visibility test (see composite_blocks)


def composite_blocks(renderer_forward, rays_o, rays_d, viewdirs, cam_origin, block_centroid, p=4.0, min_opacity=0.05,
                     group=None, ref_distance=None, all_centroids=None, **render_kwargs):
    """One block model per rank, the same rays on every rank, ONE all-reduce(sum) of [R+1,5] fp32, no host sync.

    The reference's FourierGrid path never composites blocks (it renders each block's own image subset,
    run_render.py:146-207); the only merging rule in the repository is the legacy Block-NeRF evaluation
    (eval_block_nerf.py:95-133,215-225): keep a block if its mean visibility exceeds 0.05, weight it by the inverse
    camera-to-centroid distance to the power p (IDW_Power = 4), normalise the weights, blend the images.  This
    function applies that rule to the fused renderer's float outputs: visibility of a block := its mean
    accumulated opacity 1 - alphainv_last over the rays (the FourierGrid model has no visibility network), weight
    w_b = |cam_origin - block_centroid|^-p if visible, result = sum_b w_b * [rgb, depth, alphainv_last] / sum_b w_b.

    Everything stays on the device: the visibility test is a device-side comparison folded into w_b, and row R of the
    reduced buffer carries (sum_b w_b, number of visible blocks, 0, 0, 0), so one collective returns both the numerators
    and the normaliser.  A block that FAILS the test is not dropped but scaled by 2^-60: next to any visible block its
    contribution is below fp32 resolution (the blend is bit-identical to excluding it unless its distance weight exceeds
    the visible ones' by > 10^10), and when NO block is visible -- the reference skips such a frame
    (eval_block_nerf.py:225-226) -- the common factor cancels and the frame is the inverse-distance blend of all blocks,
    without a second collective or a host decision.

    Scale of the distance weight.  Only the RATIOS of the w_b matter, and |cam - centroid|^-p itself is not a safe fp32 quantity:
    un-centred city-scale coordinates put cameras 1e4 ... 1e5 units from the centroids (1e-20, and 1e-20 * 2^-60 is denormal), a
    camera inside a block makes it huge.  The weight is therefore formed in RELATIVE space, (|cam - centroid| / ref)^-p in float64
    with ref = the distance of the NEAREST block, so the nearest block weighs exactly 1 and every other block <= 1 whatever the
    units; only the far end is clamped, at 2^-60 (a block 2^15 times farther than the nearest at p = 4: 36 binary orders below
    fp32's resolution next to it -- the clamp keeps w_b * 2^-60 a normal number, it never changes a ratio that could be seen).
    `ref` must be the same on every rank:
      * `all_centroids` (the centroids of ALL blocks, e.g. from the scene's block table): every rank computes the minimum itself --
        the function stays at ONE collective;
      * or `ref_distance` (any agreed positive number, e.g. the block spacing; weights may then exceed 1);
      * neither: one extra 8-byte all-reduce(MIN) of the rank's own distance agrees on it (device-side, no host sync).
    (Round 4 clamped the ABSOLUTE weight into [2^-40, 2^40]: with the default ref = 1 every camera further than 1024 units from
    every centroid got the same weight and the blend silently became a plain average -- ADVICE r4.)
    Returned `block_weight` is this rank's NORMALISED weight as a 0-d device tensor (w_b / sum_b w_b), `visible_blocks` the number
    of blocks that passed the test."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    kw = dict(render_kwargs)
    kw["render_depth"] = True
    out = renderer_forward(rays_o, rays_d, viewdirs, **kw)
    R = rays_o.shape[0]
    dev = rays_o.device
    cam64 = torch.as_tensor(cam_origin, dtype=torch.float64).reshape(3).cpu()
    d_own = (cam64 - torch.as_tensor(block_centroid, dtype=torch.float64).reshape(3).cpu()).norm()      # host arithmetic
    if ref_distance is not None:
        ref = torch.as_tensor(float(ref_distance), dtype=torch.float64)
    elif all_centroids is not None:
        cents = torch.stack([torch.as_tensor(c, dtype=torch.float64).reshape(3).cpu() for c in all_centroids])
        ref = (cents - cam64[None]).norm(dim=1).min()
    elif ws > 1:
        ref = d_own.to(dev).reshape(1).clone()
        dist.all_reduce(ref, op=dist.ReduceOp.MIN, group=group)          # 8 bytes; stays on the device
        ref = ref[0]
    else:
        ref = d_own
    rel = d_own.to(ref.device) / ref.clamp_min(1e-300)
    dw = (rel ** (-float(p))).clamp(2.0 ** -60, 2.0 ** 60).to(device=dev, dtype=torch.float32)  # one scalar upload at most
    # (upper end: only an explicit ref_distance can put a block above 1 -- a camera sitting ON a centroid must not produce inf)
    opacity = (1.0 - out["alphainv_last"]).mean() if R > 0 else torch.zeros((), device=dev)
    visible = (opacity > min_opacity).to(torch.float32)                  # 0-d device tensor
    w = dw * (visible + (1.0 - visible) * INVISIBLE_SCALE)
    acc = torch.empty(R + 1, 5, dtype=torch.float32, device=dev)
    acc[:R, 0:3] = out["rgb_marched"] * w
    acc[:R, 3] = out["depth"] * w
    acc[:R, 4] = out["alphainv_last"] * w
    acc[R, 0] = w
    acc[R, 1] = visible
    acc[R, 2:] = 0.0
    if ws > 1:
        dist.all_reduce(acc, group=group)
    norm = acc[R, 0].clamp_min(2.0 ** -120)
    blended = acc[:R] / norm
    return {"rgb_marched": blended[:, 0:3].contiguous(), "depth": blended[:, 3].contiguous(),
            "alphainv_last": blended[:, 4].contiguous(), "block_weight": w / norm, "visible_blocks": acc[R, 1]}
