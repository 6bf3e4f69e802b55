"""Phase profile of the shade kernel on the S1 frame (needs the -DUG_SHADE_PROF build of the library):

    UG_OUT=../../build/ab/lib_prof.so UG_SHADE_FLAGS=-DUG_SHADE_PROF bash unboundednerfpytorch_amd/csrc/build.sh
    UGRID_LIB=build/ab/lib_prof.so python tools/gpu_shade_phases.py          (GPU box)

Prints, per phase of ug_shade_tile, the shader-clock ticks summed over all waves, its share, and ticks per 32-survivor
pass per wave (s_memtime instrumentation itself costs ~10 % of the wave cycles)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ["tile set-up (embedding table, first entries)", "k0 gather round 0 (+ LDS transpose)", "k0 gather round 1",
         "layer 1 (bias, split, 36 MFMA)", "layer 2 (relu, split, 96 MFMA)", "layer 3 + sigmoid (VALU)",
         "per-ray accumulation (LDS)", "tile scheduling (atomic) + list head"]


def main():
    import bench
    from unboundednerfpytorch_amd import _lib
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_a_view
    L = _lib.load()
    L.ugx_shade_prof_read.restype = ctypes.c_int
    L.ugx_shade_prof_read.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    G, H, W = 200, 1080, 1920
    state = bench.make_state(G, dev, seed=0)
    rend = FourierGridRenderer(state, dev)
    del state
    K = [[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]]
    ro, rd, vd = get_rays_of_a_view(H, W, K, bench.camera(0, dev))
    ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in (ro, rd, vd)]
    buf = (ctypes.c_uint64 * 8)()
    rend(ro, rd, vd, stepsize=1.31, render_depth=True)
    torch.cuda.synchronize()
    L.ugx_shade_prof_read(buf)      # discard the warm-up frame
    n = 3
    timing = []
    for _ in range(n):
        rend(ro, rd, vd, stepsize=1.31, render_depth=True, timing=timing)
    torch.cuda.synchronize()
    L.ugx_shade_prof_read(buf)
    M = rend.survivors_of_last_chunk()
    ticks = [int(x) / n for x in buf]
    tot = sum(ticks)
    passes = M / 32.0
    shade_ms = sum(ev[-2].elapsed_time(ev[-1]) for ev, _ in timing) / n
    print("shade kernel %.3f ms (instrumented build); %d survivors = %.0f passes; 2048 waves; total %.3e wave-ticks"
          % (shade_ms, M, passes, tot))
    for name, t in zip(NAMES, ticks):
        print("  %-48s %5.1f %%   %8.0f ticks per pass per wave" % (name, 100.0 * t / tot, t / passes))
    print("  sum %.0f ticks per pass per wave; wall: 2 waves per SIMD -> %.0f ticks per pass per SIMD if they never overlapped"
          % (tot / passes, tot / passes))


if __name__ == "__main__":
    main()
