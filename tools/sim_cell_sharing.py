"""How many DISTINCT grid cells does one 64-ray march wave touch per (sample, level)?  (VERDICT r4 item 2 / "missing" #3)

Host-side geometry study on the S1 frame (bench.py's camera, G = 200, F = 3, stepsize 1.31 -> S = 256; or --stepsize 0.5 /
--freq 4 for the truck shape): for a sample of 8 x 8 pixel tiles in Z-order lane layout (fourier_render.pixel_tile_order) the
cell index of every (lane, sample, level) is formed as the kernel forms it, and per (tile, sample, level) the script reports

  * distinct cells per wave instruction (what an ideal de-duplicating fetch would load),
  * the fraction of lane quads (2 x 2 pixels) whose four lanes share one cell,
  * the extent of the wave's cells per axis (bounding box), and how often ALL 64 lanes fit a 2x2x2 box anchored at the wave's
    minimum cell / a 3x3x3 box centred on the cell of the tile's centre lane -- the hit rate of an LDS-staged neighbourhood,
  * the fraction of lanes that would need the fall-back global load when they do not.

No GPU needed; writes one JSON (profiles/r05/cell_sharing_<tag>.json) that DESIGN.md section 5.5 prices the alternatives with.
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def zorder_xy():
    lane = np.arange(64)
    x = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4)
    y = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4)
    return x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--freq", type=int, default=3)
    ap.add_argument("--stepsize", type=float, default=1.31)
    ap.add_argument("--tiles", type=int, default=400)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench
    from unboundednerfpytorch_amd.fourier_render import sample_table  # noqa: E402  (host-only helper; loads the library)
    H, W, G, F = 1080, 1920, a.grid, a.freq
    P = 2 * F + 1
    c2w = bench.camera(0, "cpu").double().numpy()
    K = np.array([[1600.0, 0, W / 2.0], [0, 1600.0, H / 2.0], [0, 0, 1]])
    t_tab, _ = sample_table(G, a.stepsize, 0.2)
    t = t_tab.double().numpy()
    S = t.shape[0]
    rng = np.random.default_rng(0)
    tiles_y = rng.integers(0, H // 8, a.tiles)
    tiles_x = rng.integers(0, W // 8, a.tiles)
    lx, ly = zorder_xy()
    stats = {l: {"distinct": [], "quad_uniform": [], "ext": [], "fit222": [], "fit333c": [], "miss333c_lanes": [], "wave_uniform": []}
             for l in range(P)}
    for ty, tx in zip(tiles_y, tiles_x):
        i = tx * 8 + lx + 0.5
        j = ty * 8 + ly + 0.5
        dirs = np.stack([(i - K[0, 2]) / K[0, 0], -(j - K[1, 2]) / K[1, 1], -np.ones_like(i)], -1)
        rd = dirs @ c2w[:3, :3].T
        rd = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
        ro = c2w[:3, 3][None, :]
        p = ro[:, None, :] + rd[:, None, :] * t[None, :, None]                  # [64, S, 3]
        nrm = np.abs(p).max(-1, keepdims=True)
        out = nrm > 1.0
        pc = np.where(out, p / np.maximum(nrm, 1e-30) * (1.2 - 0.2 / np.maximum(nrm, 1e-30)), p)
        u = (pc + 1.2) / 2.4 * 2.0 - 1.0                                         # [64, S, 3] in [-1, 1]
        for l in range(P):
            if l == 0:
                c = u
            else:
                k = (l - 1) // 2
                c = np.sin((2 ** k) * u) if (l - 1) % 2 == 0 else np.cos((2 ** k) * u)
            cell = np.clip(np.floor((c + 1.0) / 2.0 * (G - 1)), 0, G - 2).astype(np.int64)   # [64, S, 3]
            lin = (cell[..., 0] * (G - 1) + cell[..., 1]) * (G - 1) + cell[..., 2]           # [64, S]
            srt = np.sort(lin, axis=0)
            distinct = 1 + (np.diff(srt, axis=0) != 0).sum(0)                                 # [S]
            quads = lin.reshape(16, 4, S)
            quad_uniform = (quads == quads[:, :1]).all(1).mean(0)                             # [S]
            lo, hi = cell.min(0), cell.max(0)                                                 # [S, 3]
            ext = (hi - lo + 1)
            fit222 = (ext <= 2).all(-1)
            ctr = cell[48]                                                                    # lane 48 = pixel (4, 4)
            d = np.abs(cell - ctr[None])
            inbox = (d <= 1).all(-1)                                                          # [64, S]
            st = stats[l]
            st["distinct"].append(distinct)
            st["quad_uniform"].append(quad_uniform)
            st["ext"].append(ext.max(-1))
            st["fit222"].append(fit222)
            st["fit333c"].append(inbox.all(0))
            st["miss333c_lanes"].append(1.0 - inbox.mean(0))
            st["wave_uniform"].append(distinct == 1)
    res = {"workload": "S1 camera, 1920x1080, G=%d, F=%d, stepsize %g -> S=%d; %d random 8x8 tiles, Z-order lanes" % (G, F, a.stepsize, S, a.tiles),
           "levels": {}}
    alld = []
    for l in range(P):
        st = {k: np.stack(v) for k, v in stats[l].items()}                                   # [tiles, S]
        d = st["distinct"]
        alld.append(d)
        res["levels"][str(l)] = {
            "distinct_cells_mean": float(d.mean()), "distinct_cells_p50": float(np.median(d)), "distinct_cells_p95": float(np.percentile(d, 95)),
            "distinct_cells_max": int(d.max()), "wave_uniform_frac": float(st["wave_uniform"].mean()),
            "quad_uniform_frac_of_quads": float(st["quad_uniform"].mean()),
            "bbox_extent_max_axis_mean": float(st["ext"].mean()), "bbox_extent_max_axis_p95": float(np.percentile(st["ext"], 95)),
            "all_lanes_fit_2x2x2_at_min_frac": float(st["fit222"].mean()),
            "all_lanes_fit_3x3x3_centred_frac": float(st["fit333c"].mean()),
            "lanes_outside_3x3x3_centred_frac": float(st["miss333c_lanes"].mean()),
        }
    alld = np.stack(alld)                                                                    # [P, tiles, S]
    res["all_levels"] = {"distinct_cells_mean": float(alld.mean()),
                         "duplication_factor_64_over_distinct": float(64.0 / alld.mean()),
                         "per_sample_all_7_levels_fit_2x2x2_frac": float(np.stack([np.stack(stats[l]["fit222"]) for l in range(P)]).all(0).mean()),
                         "per_sample_all_levels_fit_3x3x3_centred_frac": float(np.stack([np.stack(stats[l]["fit333c"]) for l in range(P)]).all(0).mean())}
    # by sample index: inner (first half) vs outer (contracted) samples
    half = S // 2
    res["inner_vs_outer"] = {"inner_distinct_mean": float(alld[:, :, :half].mean()), "outer_distinct_mean": float(alld[:, :, half:].mean())}
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
