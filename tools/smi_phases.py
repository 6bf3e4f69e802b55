"""Workload for tools/smi_trace.py: the S1 frame's two kernels in labelled phases of ~3 s each -- idle, march only, shade only, whole
frames -- so that the power / clock trace can be read per kernel.  Prints `##PHASE name` before each phase and the kernels' mean time in it."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_pixel_index, pixel_tile_order
    state = bench.make_state(200, dev, seed=0)
    rend = FourierGridRenderer(state, dev)
    del state
    H, W = 1080, 1920
    K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
    ro, rd, vd = get_rays_of_pixel_index(H, W, K, bench.camera(0, dev), pixel_tile_order(H, W, dev))
    kw = dict(stepsize=1.31, render_depth=True, ray_order="coherent")
    rend(ro, rd, vd, **kw)
    torch.cuda.synchronize()
    print("##PHASE idle", flush=True)
    time.sleep(3.0)

    from unboundednerfpytorch_amd import _lib
    L, P_ = _lib.load(), _lib.ptr
    t_tab, s_tab, S = rend.tables(1.31)
    R = ro.shape[0]
    prm = rend._params(R, S, 1.31)
    ws = rend._workspace(R, S)
    last, depth, rgb = torch.empty(R, device=dev), torch.empty(R, device=dev), torch.empty(R, 3, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    def march():
        _lib.check(L.ugrid_render_march(prm, P_(ro), P_(rd), P_(t_tab), P_(s_tab), P_(rend.density_bricks), P_(last), P_(depth), P_(ws), st), "march")

    def shade():
        _lib.check(L.ugrid_render_shade(prm, P_(vd), P_(rend.k0_bricks), P_(rend.mlp_packed), P_(ws), P_(rgb), st), "shade")

    def loop(name, fns, secs=3.0):
        print("##PHASE " + name, flush=True)
        n, t0 = 0, time.time()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        while time.time() - t0 < secs:
            for _ in range(20):
                for f in fns:
                    f()
            n += 20
            torch.cuda.synchronize()
        ev1.record()
        torch.cuda.synchronize()
        print("%s: %.3f ms per iteration over %d iterations" % (name, ev0.elapsed_time(ev1) / n, n), flush=True)
    march()
    torch.cuda.synchronize()
    loop("march_only", [march])
    loop("shade_only", [shade])
    loop("frame_march_then_shade", [march, shade])
    print("##PHASE idle_after", flush=True)
    time.sleep(2.0)


if __name__ == "__main__":
    main()
