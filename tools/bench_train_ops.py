"""S3-shaped measurement of the per-voxel training passes: masked Adam, dense Adam and the
total-variation gradient on a k0-sized parameter (P=7, C=12, G=200 -> 672 M voxels, 2.69 GB per array).
Algorithmic bytes: masked Adam 4 B/voxel (grad) + 24 B per touched voxel; dense Adam 28 B/voxel; TV 7 reads + 1
write of 4 B per processed voxel.  Prints GB/s against the 8 TB/s HBM peak."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unboundednerfpytorch_amd import adam_upd_cuda, total_variation_cuda

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n): fn()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n

shape = (7, 12, 200, 200, 200)
N = 1
for s in shape: N *= s
dev = "cuda"
p = torch.randn(shape, device=dev); m = torch.zeros_like(p); v = torch.zeros_like(p)
g = torch.randn(shape, device=dev)
frac = 0.05
mask = torch.rand(shape, device=dev) < frac
gs = torch.where(mask, g, torch.zeros_like(g)); del mask
touched = int((gs != 0).sum())
res = {}
t = timed(lambda: adam_upd_cuda.masked_adam_upd(p, gs, m, v, 3, 0.9, 0.99, 0.1, 1e-8))
res["masked_adam (5% touched)"] = (t, 4 * N + 24 * touched)
t = timed(lambda: adam_upd_cuda.adam_upd(p, g, m, v, 3, 0.9, 0.99, 0.1, 1e-8))
res["adam dense"] = (t, 28 * N)
gg = g.clone()
t = timed(lambda: total_variation_cuda.total_variation_add_grad(p, gg, 1e-3, 1e-3, 1e-3, True))
res["TV dense"] = (t, 32 * N)
gsp = gs.clone()
t = timed(lambda: total_variation_cuda.total_variation_add_grad(p, gsp, 1e-3, 1e-3, 1e-3, False))
res["TV masked (5%)"] = (t, 4 * N + 32 * touched)
for k, (ms, b) in res.items():
    print("%-26s %8.3f ms  %8.1f GB/s algorithmic  (%.2f of 8 TB/s)" % (k, ms, b / ms / 1e6, b / ms / 1e6 / 8000))

# grid lookup forward / backward at the config-3 training shape: 4096 rays x 668 samples through the density grid
# (all samples), ~5 % of them through the 12-channel k0 grid; HIP op against torch's grid_sample autograd
del p, m, v, g, gs, gg, gsp
torch.cuda.empty_cache()
import torch.nn.functional as Fn
from unboundednerfpytorch_amd.grid import GridQuery
Fq, G = 3, 200
lo, hi = torch.full((3,), -1.2, device=dev), torch.full((3,), 1.2, device=dev)
def torch_query(grid, pts):
    u = ((pts - lo) / (hi - lo)).flip((-1,)) * 2 - 1
    lv = [u]
    for k in range(Fq):
        lv += [torch.sin(u * 2 ** k), torch.cos(u * 2 ** k)]
    taps = Fn.grid_sample(grid, torch.stack(lv, 0)[:, None, None], mode='bilinear', align_corners=True)
    return taps.mean(0).reshape(grid.shape[1], -1).T
for name, C, n in (("density grid (C=1), 2.74 M points", 1, 4096 * 668), ("k0 grid (C=12), 137 k points", 12, 4096 * 668 // 20)):
    grid = torch.randn(7, C, G, G, G, device=dev, requires_grad=True)
    pts = torch.rand(n, 3, device=dev) * 2.4 - 1.2
    gout = torch.randn(n, C, device=dev)
    def fb(fn):
        grid.grad = None
        out = fn().reshape(n, C)
        out.backward(gout)
    t_hip = timed(lambda: fb(lambda: GridQuery.apply(grid, pts, lo, hi, Fq)))
    t_ref = timed(lambda: fb(lambda: torch_query(grid, pts)))
    t_f = timed(lambda: GridQuery.apply(grid.detach(), pts, lo, hi, Fq))
    print("%-36s fwd+bwd HIP %7.3f ms (fwd %6.3f)   torch grid_sample fwd+bwd %7.3f ms   x%.1f" % (name, t_hip, t_f, t_ref, t_ref / t_hip))
