"""Drop-in ops vs the REFERENCE'S OWN kernels on the same MI355X: every function of the four
extension modules timed at hot-path sizes through this package's modules (libugrid_hip.so) and through the reference's
FourierGrid/cuda sources compiled for gfx950 by oracle/build_ref.py (`fma` build = the compiler's default contraction,
how the reference's own setup.py would build them).  The reference binaries are test infrastructure: this tool and the
tests are the only things that load them.

    python tools/bench_dropin_ops.py > gpurun_out/dropin_ops.txt        (GPU box; needs oracle/_ref)

Sizes: config-3 training batch (4096 rays x 668 samples, DVGO-style variable-length sampling with ~0.4 M points), a
k0-sized parameter for TV / Adam (P=7, C=12, G=200: 672 M voxels)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    from oracle import build_ref
    from unboundednerfpytorch_amd import adam_upd_cuda, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    mine = {"render_utils_cuda": render_utils_cuda, "total_variation_cuda": total_variation_cuda,
            "ub360_utils_cuda": ub360_utils_cuda, "adam_upd_cuda": adam_upd_cuda}
    ref = build_ref.load("fma") if build_ref.built("fma") else None
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    R, S = 4096, 668
    rows = []

    def both(name, mod, fn, make, note=""):
        t_m = timed(lambda: getattr(mine[mod], fn)(*make()))
        t_r = timed(lambda: getattr(ref[mod], fn)(*make())) if ref is not None else float("nan")
        rows.append((name, t_m, t_r, note))

    # ---- samplers (dvgo.py:306-328): 65 536 rays through a unit box
    Rs = 65536
    o = torch.randn(Rs, 3, device=dev, generator=g) * 0.3
    d = torch.randn(Rs, 3, device=dev, generator=g)
    lo, hi = torch.tensor([-1.0, -1.0, -1.0], device=dev), torch.tensor([1.0, 1.0, 1.0], device=dev)
    both("infer_t_minmax (65k rays)", "render_utils_cuda", "infer_t_minmax", lambda: (o, d, lo, hi, 0.2, 1e9))
    tmin, tmax = render_utils_cuda.infer_t_minmax(o, d, lo, hi, 0.2, 1e9)
    both("infer_n_samples", "render_utils_cuda", "infer_n_samples", lambda: (d, tmin, tmax, 0.005))
    both("infer_ray_start_dir", "render_utils_cuda", "infer_ray_start_dir", lambda: (o, d, tmin))
    out = render_utils_cuda.sample_pts_on_rays(o, d, lo, hi, 0.2, 1e9, 0.005)
    both("sample_pts_on_rays (%.1f M points)" % (out[0].shape[0] / 1e6), "render_utils_cuda", "sample_pts_on_rays",
         lambda: (o, d, lo, hi, 0.2, 1e9, 0.005))
    both("sample_ndc_pts_on_rays (65k x 128)", "render_utils_cuda", "sample_ndc_pts_on_rays", lambda: (o, d, lo, hi, 128))
    both("sample_bg_pts_on_rays (65k x 128)", "render_utils_cuda", "sample_bg_pts_on_rays", lambda: (o, d, tmax, 0.3, 128))
    pts = out[0]
    world = torch.rand(200, 200, 200, device=dev, generator=g) > 0.5
    sc = torch.tensor([99.5, 99.5, 99.5], device=dev)
    sh = torch.tensor([99.5, 99.5, 99.5], device=dev)
    both("maskcache_lookup (%.1f M points, 200^3 mask)" % (pts.shape[0] / 1e6), "render_utils_cuda", "maskcache_lookup",
         lambda: (world, pts, sc, sh))
    # ---- per-sample ops at the config-3 batch: 4096 rays x 668 samples
    n = R * S
    dens = torch.randn(n, device=dev, generator=g) * 6 + 2
    both("raw2alpha (2.7 M)", "render_utils_cuda", "raw2alpha", lambda: (dens, -9.21, 0.5))
    e, alpha = render_utils_cuda.raw2alpha(dens, -9.21, 0.5)
    gb = torch.randn(n, device=dev, generator=g)
    both("raw2alpha_backward", "render_utils_cuda", "raw2alpha_backward", lambda: (e, gb, 0.5))
    itv = torch.rand(n, device=dev, generator=g) + 0.1
    both("raw2alpha_nonuni", "render_utils_cuda", "raw2alpha_nonuni", lambda: (dens, -9.21, itv))
    both("raw2alpha_nonuni_backward", "render_utils_cuda", "raw2alpha_nonuni_backward", lambda: (e, gb, itv))
    a_small = torch.rand(n, device=dev, generator=g) * 0.02
    ray_id = torch.arange(R, device=dev).repeat_interleave(S)
    both("alpha2weight (4096 rays x 668)", "render_utils_cuda", "alpha2weight", lambda: (a_small, ray_id, R),
         "reference: one thread per ray, serial; here: one wave per ray, coalesced, same serial chain bit for bit")
    w, T, last, i_s, i_e = render_utils_cuda.alpha2weight(a_small, ray_id, R)
    gw, gl = torch.randn(n, device=dev, generator=g), torch.randn(R, device=dev, generator=g)
    both("alpha2weight_backward", "render_utils_cuda", "alpha2weight_backward", lambda: (a_small, w, T, last, i_s, i_e, R, gw, gl))
    dist = torch.rand(8192, 667, device=dev, generator=g) * 0.02
    both("cumdist_thres (8192 x 667)", "ub360_utils_cuda", "cumdist_thres", lambda: (dist, 0.0114))
    # ---- k0-sized parameter passes
    shape = (7, 12, 200, 200, 200)
    p = torch.randn(shape, device=dev, generator=g)
    gr = torch.randn(shape, device=dev, generator=g)
    gs = torch.where(torch.rand(shape, device=dev, generator=g) < 0.05, gr, torch.zeros_like(gr))
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    both("total_variation_add_grad dense (672 M voxels)", "total_variation_cuda", "total_variation_add_grad",
         lambda: (p, gr, 1e-3, 1e-3, 1e-3, True))
    both("total_variation_add_grad masked (5 %)", "total_variation_cuda", "total_variation_add_grad",
         lambda: (p, gs, 1e-3, 1e-3, 1e-3, False))
    both("adam_upd (672 M)", "adam_upd_cuda", "adam_upd", lambda: (p, gr, m, v, 3, 0.9, 0.99, 0.1, 1e-8))
    both("masked_adam_upd (5 % touched)", "adam_upd_cuda", "masked_adam_upd", lambda: (p, gs, m, v, 3, 0.9, 0.99, 0.1, 1e-8))
    lr = torch.rand(shape, device=dev, generator=g)
    both("adam_upd_with_perlr", "adam_upd_cuda", "adam_upd_with_perlr", lambda: (p, gr, m, v, lr, 3, 0.9, 0.99, 0.1, 1e-8))
    alt = torch.empty_like(p)
    t_f = timed(lambda: adam_upd_cuda.tv_adam_dense(p, alt, gr, m, v, 1e-3, 1e-3, 1e-3, 3, 0.9, 0.99, 0.1, 1e-8, True))
    rows.append(("dense TV + masked Adam, FUSED (new entry point)", t_f, float("nan"), "the two reference calls above it: dense TV + adam"))
    print("%-52s %12s %14s %8s" % ("op (sizes)", "this repo ms", "reference ms", "speedup"))
    for name, t_m, t_r, note in rows:
        sp = ("%.2fx" % (t_r / t_m)) if t_r == t_r else "-"
        print("%-52s %12.3f %14s %8s  %s" % (name, t_m, ("%.3f" % t_r) if t_r == t_r else "-", sp, note))


if __name__ == "__main__":
    main()
