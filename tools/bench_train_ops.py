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
