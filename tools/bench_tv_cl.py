"""Fused dense TV + Adam on the S3 k0 array (9 x 12 x 200^3), channel-last, channel-last and canonical, with the linear and the XCD-contiguous
block order and non-temporal streams (ugrid_tune('tv_xcd') 0 | 1 | 2).  Prints ms per launch and the 7-pass bandwidth."""
import json
import sys

import torch

sys.path.insert(0, ".")
from unboundednerfpytorch_amd import adam_upd_cuda, fourier_render  # noqa: E402


def time_it(fn, n=6):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    shape = (9, 12, 200, 200, 200)
    dev = torch.device("cuda", 0)
    res = {}
    for layout in ("channels_last", "canonical"):
        mk = (lambda: torch.empty(shape, device=dev).contiguous(memory_format=torch.channels_last_3d)) if layout == "channels_last" \
            else (lambda: torch.empty(shape, device=dev))
        p, out, g, m, v = mk().normal_(), mk(), mk().normal_(), mk().zero_(), mk().zero_()
        for mode in (0, 1, 2):
            fourier_render.tune("tv_xcd", mode)
            ms = time_it(lambda: adam_upd_cuda.tv_adam_dense(p, out, g, m, v, 0.1, 0.1, 0.1, 3, 0.9, 0.99, 0.1, 1e-8, True))
            res["%s/xcd%d" % (layout, mode)] = {"ms": ms, "TBps_7pass": 7 * p.numel() * 4 / ms / 1e9}
            from unboundednerfpytorch_amd import total_variation_cuda
            ms = time_it(lambda: total_variation_cuda.total_variation_add_grad(p, g, 0.1, 0.1, 0.1, True))
            res["%s/xcd%d" % (layout, mode)].update({"tv_only_ms": ms, "tv_only_TBps_3pass": 3 * p.numel() * 4 / ms / 1e9})
        del p, out, g, m, v
    fourier_render.tune("tv_xcd", 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
