"""The ray-order guard of FourierGridRenderer.forward on the S1 frame: the frame's rays in 8 x 8 pixel blocks (what
render_view does), shuffled with ray_order="coherent" (rendered as given: the 8.6x cliff of profiles/r02), and shuffled with
the default ray_order="auto" (detected, Morton-sorted on the device, results put back).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import warnings
    import bench
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_pixel_index, pixel_tile_order
    dev = torch.device("cuda", 0)
    G, H, W = 200, 1080, 1920
    rend = FourierGridRenderer(bench.make_state(G, dev, seed=0), dev)
    K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
    ro, rd, vd = get_rays_of_pixel_index(H, W, K, bench.camera(0, dev), pixel_tile_order(H, W, dev))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    perm = torch.randperm(ro.shape[0], device=dev, generator=g)
    so, sd, sv = ro[perm].contiguous(), rd[perm].contiguous(), vd[perm].contiguous()

    def timed(o, d, v, **kw):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = rend(o, d, v, stepsize=1.31, render_depth=True, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = rend(o, d, v, stepsize=1.31, render_depth=True, **kw)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 * 1e3, out

    t_blocks, ref = timed(ro, rd, vd, ray_order="coherent")
    t_shuf, a = timed(so, sd, sv, ray_order="coherent")
    t_auto, b = timed(so, sd, sv)
    same = all(torch.equal(ref[k][perm], a[k]) and torch.equal(a[k], b[k]) for k in ("rgb_marched", "depth", "alphainv_last"))
    print(json.dumps({"workload": "S1 frame, 2 073 600 rays", "ms_pixel_blocks": t_blocks, "ms_shuffled_rendered_as_given": t_shuf,
                      "ms_shuffled_ray_order_auto": t_auto, "tile_spread_blocks": float(rend.tile_spread(ro, vd)),
                      "tile_spread_shuffled": float(rend.tile_spread(so, sv)), "results_bitwise_equal": bool(same)}))


if __name__ == "__main__":
    main()
