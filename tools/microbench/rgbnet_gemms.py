"""Times the rgbnet GEMMs of one S3 train step (M = 84k survivors, fp32, rocBLAS through torch) one by one, and the
split-K alternatives for the skinny ones.  Prints us per call."""
import json

import torch


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def splitk(go, x, S):
    M = go.shape[0]
    m = M // S
    main = m * S
    out = torch.bmm(go[:main].view(S, m, -1).transpose(1, 2), x[:main].view(S, m, -1)).sum(0)
    if main < M:
        out = out + go[main:].t() @ x[main:]
    return out


def main():
    dev = torch.device("cuda", 0)
    M = 84123
    res = {}
    for name, K, N in (("l1", 39, 128), ("l2", 128, 128), ("l3", 128, 3)):
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev)
        b = torch.randn(N, device=dev)
        go = torch.randn(M, N, device=dev)
        res[name + "_fwd_addmm"] = t(lambda: torch.addmm(b, x, W.t()))
        res[name + "_gx"] = t(lambda: go @ W)
        res[name + "_gW"] = t(lambda: go.t() @ x)
        for S in (32, 128, 512):
            res[name + "_gW_splitk%d" % S] = t(lambda: splitk(go, x, S))
        ref = go.t() @ x
        res[name + "_gW_splitk128_relerr"] = float((splitk(go, x, 128) - ref).abs().max() / ref.abs().max())
        res[name + "_gb"] = t(lambda: go.sum(0))
    # the N = 3 layer by broadcasting instead of a GEMM
    x = torch.randn(M, 128, device=dev); W = torch.randn(3, 128, device=dev); go = torch.randn(M, 3, device=dev)
    res["l3_fwd_einsum"] = t(lambda: (x.unsqueeze(1) * W.unsqueeze(0)).sum(-1))
    res["l3_gx_bcast"] = t(lambda: go[:, 0:1] * W[0] + go[:, 1:2] * W[1] + go[:, 2:3] * W[2])
    print(json.dumps({k: round(v, 6) if v < 1 else round(v, 1) for k, v in res.items()}))


if __name__ == "__main__":
    main()
