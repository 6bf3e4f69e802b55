"""CI stand-in with FourierGridRenderer's call signature (CPU, torch ops): per-ray outputs are a deterministic function of
the ray alone, so bench.py's launch / sharding / exchange / frame-assembly logic can be checked without a GPU
(UGRID_BENCH_STANDIN=bench_standin:Renderer, tests/test_host_logic.py).  Not a renderer, not a fallback: bench.py marks
such a run `"renderer": "stand-in"` and emits no roofline / cpu_baseline."""
import torch


class _Event:
    def elapsed_time(self, other):
        return 0.0


class Renderer:
    pipeline = 0

    def tables(self, stepsize):
        return None, None, 16

    def rays_per_chunk(self, S):
        return 1 << 30

    def survivors_of_last_chunk(self):
        return 0

    def __call__(self, ro, rd, vd, stepsize=None, render_depth=True, timing=None):
        if timing is not None:
            timing.append(((_Event(), _Event(), _Event()), ro.shape[0]))
        return {"rgb_marched": torch.sin(vd * 3.0), "depth": (rd * vd).sum(-1), "alphainv_last": torch.cos(vd[:, 0] * 5.0), "n_max": 16}
