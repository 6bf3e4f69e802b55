"""The fused dense TV + Adam pass on the S3 k0 array (P = 9, C = 12, G = 200, channel-last, 3.46 GB) with and without the
touched-line bitmap of the gradient (6 % of the lines marked, like a training step): time per pass."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unboundednerfpytorch_amd import _lib, adam_upd_cuda
L = _lib.load()
P, C, G = 9, 12, 200
dev = "cuda"
p = torch.randn(P, C, G, G, G, device=dev).contiguous(memory_format=torch.channels_last_3d)
po = torch.empty_like(p, memory_format=torch.preserve_format)
g = torch.zeros_like(p, memory_format=torch.preserve_format)
m = torch.zeros_like(p, memory_format=torch.preserve_format)
v = torch.zeros_like(p, memory_format=torch.preserve_format)
words = int(L.ugrid_touch_words(p.numel()))
for frac in (0.06, 1.0, 0.0):
    bits = (torch.rand(words * 32, device=dev) < frac)
    w = (bits.view(words, 32).long() << torch.arange(32, device=dev)).sum(1)
    touch = (w & 0xFFFFFFFF).to(torch.int64)
    touch = torch.where(touch >= 2 ** 31, touch - 2 ** 32, touch).to(torch.int32)
    for use in (False, True):
        ts = []
        for it in range(6):
            t = touch.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            adam_upd_cuda.tv_adam_dense(p, po, g, m, v, 1e-6, 1e-6, 1e-6, 3, 0.9, 0.99, 0.1, 1e-8, True, rezero_grad=True,
                                        **({"touch": t} if use else {}))
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("marked %.2f touch=%s: %.3f ms (min of 5)" % (frac, use, min(ts[1:])))
