"""Frame loop of the render driver: the MI355X counterpart of
run_render.render_viewpoints (/root/reference/FourierGrid/run_render.py:14-114).

The reference generates rays on the host, renders 8192-ray chunks through a dict-per-chunk loop, concatenates, and
calls `.cpu().numpy()` three times per frame (a blocking device sync each).  Here a frame is: rays generated on the
device, ONE fused march + shade pass over all H*W rays (sharded over the process group when one is initialised,
dist.render_sharded), and ONE asynchronous device-to-host copy of the packed [H,W,5] result into pinned memory on
a side stream, overlapped with the next frame's render.  The views of a list are independent frames: consecutive ones are
issued on two alternating streams, each with a work list of its own (frames_in_flight = 2), so that the march of view k + 1
runs beside the shade of view k -- whole frames at 1080p: 8.8 -> 8.4 ms per view on white-noise grids, 14.0 -> 12.4 ms on
a trained-like truck-shaped scene, every frame bit-identical (profiles/r06/frame_pair_n1.txt).  Same return values as the
reference: numpy arrays rgbs [N,H,W,3], depths [N,H,W,1], bgmaps [N,H,W,1] (+ PSNRs when ground truth is given)."""
import numpy as np
import torch


@torch.no_grad()
def render_viewpoints(model, render_poses, HW, Ks, render_kwargs, gt_imgs=None, render_factor=0,
                      flip_x=False, flip_y=False, group=None, verbose=False, frames_in_flight=None):
    """model: FourierGridRenderer, or a DirectVoxGORenderer / DirectContractedVoxGORenderer (their render_view takes the
    reference's render_kwargs 'near', 'far', 'bg' as well); render_poses [N,3or4,4] camera-to-world; HW [N,2]; Ks [N,3,3];
    render_kwargs: needs 'stepsize', may carry 'inverse_y' (the keys run_render.py passes; others are ignored).
    frames_in_flight (default: the model's own `frames_in_flight` attribute, else 2): n >= 2 = consecutive views take n streams and
    n work lists in turn (renderers with use_workspace_slot; every further work list costs up to 8.4 GB at 1080p x 256 samples),
    1 = one stream.  2 is within 2 % of the best for 1080p frames; a view that leaves most of the chip idle gains from 4
    (DirectVoxGO, 800 x 800: 1.88 / 1.03 / 0.78 ms per view at 1 / 2 / 4; profiles/r06/frames_in_flight_sweep.txt).
    Returns (rgbs, depths, bgmaps) or (rgbs, depths, bgmaps, psnrs) when gt_imgs is given."""
    assert len(render_poses) == len(HW) and len(HW) == len(Ks)
    HW = np.asarray(HW).copy()
    Ks = np.asarray(Ks, dtype=np.float64).copy()
    if render_factor != 0:                                    # run_render.py:22-26
        HW = (HW / render_factor).astype(int)
        Ks[:, :2, :3] /= render_factor
    dev = model.device
    if frames_in_flight is None:
        frames_in_flight = int(getattr(model, "frames_in_flight", 2))
    copy_stream = torch.cuda.Stream(dev)
    caller = torch.cuda.current_stream(dev)
    pair = None
    if (frames_in_flight >= 2 and hasattr(model, "use_workspace_slot") and len(render_poses) > 1
            and model.use_workspace_slot(0) is not False):      # (False: a model outside the fused shapes, composed forward, one stream)
        pair = [torch.cuda.Stream(dev) for _ in range(min(int(frames_in_flight), len(render_poses)))]
    n = len(render_poses)
    n_slots = len(pair) if pair is not None else 2
    host = [None] * n_slots       # pinned buffers (one per view in flight), re-allocated when the frame size changes
    done = [None] * n_slots
    frames = []

    def drain(slot, H, W):
        done[slot].synchronize()
        a = host[slot][: H * W * 5].numpy().reshape(H, W, 5).copy()
        frames.append(a)

    pending = []                  # (slot, H, W) of copies in flight, oldest first
    for i in range(n):
        H, W = int(HW[i][0]), int(HW[i][1])
        stream = caller
        if pair is not None:
            stream = pair[i % n_slots]
            model.use_workspace_slot(i % n_slots)
            if i < n_slots:
                stream.wait_stream(caller)        # (whatever prepared the model -- bricks, weight images -- ran on the caller's stream)
        with torch.cuda.stream(stream):
            packed = _render_packed(model, H, W, Ks[i], render_poses[i], render_kwargs, flip_x, flip_y, group)
            ready = stream.record_event()
        slot = i % n_slots
        if len(pending) == n_slots:     # the buffer about to be re-used must have been read out
            drain(*pending.pop(0))
        if host[slot] is None or host[slot].numel() < packed.numel():
            host[slot] = torch.empty(packed.numel(), dtype=torch.float32, pin_memory=True)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready)
            host[slot][: packed.numel()].copy_(packed, non_blocking=True)
            packed.record_stream(copy_stream)
            done[slot] = copy_stream.record_event()
        pending.append((slot, H, W))
        if verbose:
            print("render_viewpoints: frame %d/%d queued (%dx%d)" % (i + 1, n, W, H))
    while pending:
        drain(*pending.pop(0))
    if pair is not None:
        model.use_workspace_slot(0)
        for st in pair:
            caller.wait_stream(st)
    return _stack(frames, gt_imgs, render_factor, n)


def _render_packed(model, H, W, K, c2w, render_kwargs, flip_x, flip_y, group):
    """one view on the current stream -> flat [H*W*5] = rgb(3), depth, alphainv_last per pixel"""
    if hasattr(model, "fused_supported"):          # bounded / contracted VoxGO renderers: dict of per-ray outputs
        kw = {k: render_kwargs[k] for k in ("near", "far", "stepsize", "bg") if k in render_kwargs}
        out = model.render_view(H, W, K, c2w, inverse_y=bool(render_kwargs.get("inverse_y", False)),
                                flip_x=flip_x, flip_y=flip_y, render_depth=True, **kw)
        rgb, depth, bg = out["rgb_marched"], out["depth"], out["alphainv_last"]
    else:
        rgb, depth, bg = model.render_view(H, W, K, c2w, render_kwargs["stepsize"],
                                           inverse_y=bool(render_kwargs.get("inverse_y", False)),
                                           flip_x=flip_x, flip_y=flip_y, group=group)
    return torch.cat([rgb.reshape(-1, 3), depth.reshape(-1, 1), bg.reshape(-1, 1)], dim=1).reshape(-1)


def _stack(frames, gt_imgs, render_factor, n):
    rgbs = np.array([f[..., 0:3] for f in frames])
    depths = np.array([f[..., 3:4] for f in frames])
    bgmaps = np.array([f[..., 4:5] for f in frames])
    if gt_imgs is None:
        return rgbs, depths, bgmaps
    psnrs = []
    for i in range(n):
        gt = np.asarray(gt_imgs[i])
        if render_factor != 0:
            raise ValueError("ground-truth comparison needs render_factor == 0 (run_render.py:74 compares full frames)")
        psnrs.append(-10.0 * np.log10(np.mean(np.square(rgbs[i] - gt))))          # run_render.py:75
    return rgbs, depths, bgmaps, psnrs
