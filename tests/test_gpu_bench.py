"""bench.py's N > 1 path rehearsed on ONE GPU (two ranks share the card, gloo for the barrier / max-reduce / gather): the driver launches
exactly this command line with RCCL on an 8-GPU node, which this container and the 1-GPU test box do not have."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_strong_scaling_ragged():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`: one JSON line from rank 0, strong scaling (5 videos = shards
    of 3 + 2, the short one padded for the gather), the gathered latents of all ranks on rank 0, max-over-ranks timing."""
    env = dict(os.environ, VLG_BENCH_ONE_DEVICE="1", VLG_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "5", "--new-tokens", "48",
           "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[:6000]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 1 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 5 and d["config"]["shard_sizes"] == [3, 2]
    assert d["gathered_shape"] == [6, 48, 8]                 # 2 ranks x the largest shard (3) x 48 tokens x vae_embed_dim 8
    assert d["value"] > 0 and abs(d["value"] - 5 * 48 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["tokens_per_s_per_gpu"] == pytest.approx(d["value"] / 2)
    proj = d["config"]["projected_strong_speedup_from_one_gpu_shards"]      # measured shard timings from the tracked file, or None without it
    assert proj is None or ("speedup_vs_one_gpu" in proj and "s_per_step" in proj)
