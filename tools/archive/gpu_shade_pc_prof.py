"""Where the producer / consumer shade kernel spends its time on the S1 frame (needs a -DUG_SHADE_PROF build):

    UG_OUT=../../build/ab/lib_pc_prof.so UG_SHADE_FLAGS=-DUG_SHADE_PROF bash unboundednerfpytorch_amd/csrc/build.sh
    UGRID_LIB=build/ab/lib_pc_prof.so python tools/gpu_shade_pc_prof.py            (GPU box)

Per 32-survivor pass and per wave, in shader-clock ticks (s_memtime): the producers' gather and their wait for a free ring
slot, the consumers' rgbnet pass, their wait for a filled slot and their per-tile work (embedding table, result store);
then the same frame with the producers' loads switched off (consumer-bound time) and with the consumers' rgbnet switched
off (producer-bound time).  The instrumentation itself costs a few per cent."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    from unboundednerfpytorch_amd import _lib
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_pixel_index, pixel_tile_order
    L = _lib.load()
    L.ugx_shade_prof_read.restype = ctypes.c_int
    L.ugx_shade_prof_read.argtypes = [ctypes.c_void_p]
    L.ugx_shade_prof2_read.restype = ctypes.c_int
    L.ugx_shade_prof2_read.argtypes = [ctypes.c_void_p]
    L.ugx_pc_dbg_set.restype = ctypes.c_int
    L.ugx_pc_dbg_set.argtypes = [ctypes.c_int]
    dev = torch.device("cuda", 0)
    G, H, W = 200, 1080, 1920
    scene = sys.argv[1] if len(sys.argv) > 1 else "s1"
    pc = int(sys.argv[2]) if len(sys.argv) > 2 else 2           # 2 = 12-wave geometry (6 + 6 waves per CU), 1 = 8-wave (4 + 4)
    from unboundednerfpytorch_amd.fourier_render import tune
    tune("shade_pc", pc)
    n_pairs = 256 * (6 if pc == 2 else 4)
    state = (bench.make_state if scene == "s1" else bench.make_state_surfaces)(G, dev, seed=0)
    rend = FourierGridRenderer(state, dev)
    del state
    K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
    ro, rd, vd = get_rays_of_pixel_index(H, W, K, bench.camera(0, dev), pixel_tile_order(H, W, dev))
    buf = (ctypes.c_uint64 * 8)()
    buf2 = (ctypes.c_uint64 * 8)()
    n = 3
    print("scene %s, lib %s, shade_pc=%d: %d producer + %d consumer waves" % (scene, os.environ.get("UGRID_LIB", "default"), pc, n_pairs, n_pairs))
    for dbg, what in ((0, "normal"), (1, "producers skip the k0 loads (consumer-bound)"), (2, "consumers skip the rgbnet (producer-bound)"),
                      (3, "both skipped (hand-off + scheduling only)")):
        L.ugx_pc_dbg_set(dbg)
        rend(ro, rd, vd, stepsize=1.31, render_depth=True, ray_order="coherent")
        torch.cuda.synchronize()
        L.ugx_shade_prof_read(buf)      # discard the warm-up frame
        L.ugx_shade_prof2_read(buf2)
        timing = []
        for _ in range(n):
            rend(ro, rd, vd, stepsize=1.31, render_depth=True, timing=timing, ray_order="coherent")
        torch.cuda.synchronize()
        L.ugx_shade_prof_read(buf)
        L.ugx_shade_prof2_read(buf2)
        M = rend.survivors_of_last_chunk()
        v = [int(x) / n for x in buf]
        v2 = [int(x) / n for x in buf2]
        passes = v[2]
        shade_ms = sum(ev[-2].elapsed_time(ev[-1]) for ev, _ in timing) / n
        march_ms = sum(ev[0].elapsed_time(ev[1]) for ev, _ in timing) / n
        print("== %s: shade %.3f ms (march %.3f), %d survivors, %.0f passes" % (what, shade_ms, march_ms, M, passes))
        print("   producer per pass: gather %6.0f  wait-for-slot %6.0f  total %6.0f" % (v[0] / passes, v[1] / passes, v[3] / passes))
        print("   consumer per pass: rgbnet %6.0f  wait-for-data %6.0f  per-tile %6.0f  total %6.0f" % (v[4] / passes, v[5] / passes, v[7] / passes, v[6] / passes))
        print("   rgbnet phases per pass: layer 1 %6.0f  layer 2 %6.0f  layer 3 + sigmoid %6.0f  accumulation %6.0f" % (
            v2[3] / passes, v2[4] / passes, v2[5] / passes, v2[6] / passes))
        if shade_ms > 0:
            print("   effective clock %.2f GHz (consumer ticks / kernel time per wave)" % (v[6] / float(n_pairs) / (shade_ms * 1e-3) / 1e9))
    L.ugx_pc_dbg_set(0)


if __name__ == "__main__":
    main()
