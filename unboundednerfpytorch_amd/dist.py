"""Multi-GPU render: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Rays are independent and the grids + rgbnet are read-only at render time, so the
data path is: every rank holds a full replica of the bricks, renders its contiguous ray shard with the
fused kernels, and ONE all-gather exchanges the rendered tiles [rgb(3), depth, alphainv_last] = 20 B/ray.
There is no collective inside the march/shade kernels.
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


def render_sharded(renderer_forward, rays_o, rays_d, viewdirs, group=None, **render_kwargs):
    """Render rays [R,3] split across the process group; every rank returns the full
    {'rgb_marched','depth','alphainv_last'}.  `renderer_forward(o,d,v,**kw)` is FourierGridRenderer.forward
    (or any callable with the reference forward's signature and return keys)."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rk = dist.get_rank(group) if dist.is_initialized() else 0
    R = rays_o.shape[0]
    b, e = shard_bounds(R, ws, rk)
    kw = dict(render_kwargs)
    kw["render_depth"] = True
    out = renderer_forward(rays_o[b:e].contiguous(), rays_d[b:e].contiguous(), viewdirs[b:e].contiguous(), **kw)
    per = shard_bounds(R, ws, 0)[1]
    tile = torch.zeros(per, 5, dtype=torch.float32, device=rays_o.device)
    if e > b:
        tile[: e - b, 0:3] = out["rgb_marched"]
        tile[: e - b, 3] = out["depth"]
        tile[: e - b, 4] = out["alphainv_last"]
    if ws > 1:
        full = torch.empty(ws * per, 5, dtype=torch.float32, device=rays_o.device)
        dist.all_gather_into_tensor(full, tile, group=group)
    else:
        full = tile
    full = full[:R]
    return {"rgb_marched": full[:, 0:3].contiguous(), "depth": full[:, 3].contiguous(),
            "alphainv_last": full[:, 4].contiguous()}
