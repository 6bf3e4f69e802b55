"""BASELINE.json configs[4] (S5) with MORE THAN ONE real block (VERDICT r2 "what's missing" #1): two FourierGrid block
models (different seeds, different centroids) as two gloo ranks sharing this box's one GPU, composited by
dist.composite_blocks with ONE all-reduce, against a single-process evaluation of the repository's only block-merging
rule (eval_block_nerf.py:95-133,215-225) over the same two renders.  Reduced block size (the full G = 300 run is
tools/bench_s5_blocks.py --check, profiles/r03/s5_2blocks_shared_gpu.json)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_real_blocks_composited_over_gloo_equal_the_single_process_rule():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["UGRID_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "tools", "bench_s5_blocks.py"),
           "--grid", "96", "--height", "128", "--width", "256", "--steps", "1", "--check"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    chk = res["check_vs_single_process_rule"]
    print(json.dumps(chk))
    assert res["n_gpus"] == 2 and res["finite"] and chk["ok"], chk
    assert all(chk["visible"]) and abs(sum(chk["weights_normalised"]) - 1.0) < 1e-12
    assert chk["blocks_differ_linf_rgb_marched"] > 1e-2          # two genuinely different blocks were merged
