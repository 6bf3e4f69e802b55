"""Time the experimental k0-gather variants (tools/experiments/ugrid_gather_exp.hip, symbols ugx_*) on the real S1 work list.
GPU box only:  python tools/gpu_gather_variants.py [--grid 200] [--reps 3] > gpurun_out/gather_variants.txt

For every variant: ms per launch (HIP events, best and mean of `reps`), bit-checksum of the produced features against
the round-1 pair layout (quad variants must match bit for bit; vertex = grid_sample's corner sum, compared in
sum / sum of squares)."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VARIANTS = [
    # id, layout (0 pair bricks, 1 quad bricks, 2 vertex), description
    (0, 0, "pair  : lanes l/l+32 per survivor, 12 dwordx4 per lane+level, compiler-scheduled (round 1)"),
    (1, 0, "pair  : same, every lane reads lane 0's cell (diagnostic)"),
    (10, 1, "quad  : redundant set-up, 2 levels in flight, 4 waves/SIMD"),
    (11, 1, "quad  : shared set-up,    2 levels in flight, 4 waves/SIMD"),
    (12, 1, "quad  : shared set-up, every quad reads quad 0's cell (diagnostic)"),
    (13, 1, "quad  : shared set-up,    4 levels in flight, 3 waves/SIMD"),
    (14, 1, "quad  : shared set-up,    1 level  in flight, 6 waves/SIMD"),
    (15, 1, "quad  : shared set-up,    7 levels in flight, 2 waves/SIMD"),
    (16, 1, "quad  : shared set-up,    2 levels in flight, 5 waves/SIMD"),
    (20, 2, "vertex: [P][X][Y][Z][C], 8 dwordx3 per level, 1 level  in flight, 4 waves/SIMD"),
    (21, 2, "vertex:                                       2 levels in flight, 4 waves/SIMD"),
    (22, 2, "vertex:                                       3 levels in flight, 3 waves/SIMD"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    args = ap.parse_args()
    import bench
    from unboundednerfpytorch_amd import _lib
    from unboundednerfpytorch_amd.fourier_render import FourierGridRenderer, get_rays_of_a_view
    dev = torch.device("cuda", 0)
    L = _lib.load()
    L.ugx_pack_bytes.restype = ctypes.c_int64
    L.ugx_pack_bytes.argtypes = [ctypes.c_int] * 6
    L.ugx_pack.restype = ctypes.c_int
    L.ugx_pack.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]
    L.ugx_gather.restype = ctypes.c_int
    L.ugx_gather.argtypes = [ctypes.POINTER(_lib.RenderParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.ugx_feat_checksum.restype = ctypes.c_int
    L.ugx_feat_checksum.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.ugx_feat_bytes.restype = ctypes.c_int64
    L.ugx_feat_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]

    G, H, W = args.grid, args.height, args.width
    stepsize = 1.31 * G / 200.0 if G != 200 else 1.31
    state = bench.make_state(G, dev, seed=0)
    kg = state["k0_grid"]
    rend = FourierGridRenderer(state, dev)
    K = [[1600.0 * W / 1920.0, 0, W / 2.0], [0, 1600.0 * W / 1920.0, H / 2.0], [0, 0, 1]]
    ro, rd, vd = get_rays_of_a_view(H, W, K, bench.camera(0, dev))
    ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in (ro, rd, vd)]
    R = ro.shape[0]
    S = rend.tables(stepsize)[2]
    rend(ro, rd, vd, stepsize=stepsize, render_depth=True)       # fills the work list (march) and shades once
    torch.cuda.synchronize()
    M = rend.survivors_of_last_chunk()
    assert rend.rays_per_chunk(S) >= R, "frame must fit one chunk for this tool"
    p = rend._params(R, S, stepsize)
    st = torch.cuda.current_stream(dev).cuda_stream
    P, C = 7, 12
    featbuf = torch.empty(L.ugx_feat_bytes(R, S) // 4, dtype=torch.float32, device=dev)   # [tile][64*S][12], sparse use
    bricks = {1: rend.k0_bricks}       # the product layout (C == 12) is the quad layout
    for layout in (0, 2):
        n = L.ugx_pack_bytes(P, C, G, G, G, layout) // 4
        bricks[layout] = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.check(L.ugx_pack(kg.data_ptr(), P, C, G, G, G, layout, bricks[layout].data_ptr(), st), "ugx_pack")
    torch.cuda.synchronize()
    print("S1 work list: R=%d S=%d survivors M=%d (%.2f %%); k0 arrays: pair %.1f GB, quad %.1f GB, vertex %.1f GB"
          % (R, S, M, 100.0 * M / (R * S), bricks[0].numel() * 4e-9, bricks[1].numel() * 4e-9, bricks[2].numel() * 4e-9))
    chk = torch.zeros(3, dtype=torch.int64, device=dev)
    ref_bits = None
    ref_sum = None
    for vid, layout, desc in VARIANTS:
        times = []
        err = 0
        for _ in range(args.reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            err = L.ugx_gather(ctypes.byref(p), bricks[layout].data_ptr(), rend._ws.data_ptr(), featbuf.data_ptr(), vid, st)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        if err:
            print("%3d  ERROR %d   %s" % (vid, err, desc))
            continue
        times = times[1:]
        _lib.check(L.ugx_feat_checksum(rend._ws.data_ptr(), featbuf.data_ptr(), R, S, chk.data_ptr(), st), "checksum")
        torch.cuda.synchronize()
        bits = int(chk[0].item())
        sums = chk[1:].view(torch.float64).cpu().tolist()
        if vid == 0:
            ref_bits, ref_sum = bits, sums
        verdict = "bit-identical to pair" if bits == ref_bits else (
            "sum %.9g vs %.9g, sumsq %.9g vs %.9g" % (sums[0], ref_sum[0], sums[1], ref_sum[1]))
        if vid in (1, 12):
            verdict = "(diagnostic: results not comparable)"
        print("%3d  best %.3f ms  mean %.3f ms  -> %.2f TB/s of algorithmic k0 bytes (M x 2688 B)   %s\n       %s"
              % (vid, min(times), sum(times) / len(times), M * 2688 / (min(times) * 1e-3) / 1e12, desc, verdict))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
