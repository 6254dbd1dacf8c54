"""world_size-2 gloo test of the batch-shard + final-gather path on CPU."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import video_llamagen_amd  # noqa: F401
    from video_llamagen_amd import dist as vd
    r, w, _ = vd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cond = torch.arange(n_total, dtype=torch.int64)
    extra = torch.arange(n_total * 3, dtype=torch.float32).view(n_total, 3)

    def fake_generate(c, e):          # stands in for generate(): per-sample independent, deterministic
        return torch.stack([c * 10 + k for k in range(5)], 1).to(torch.int32) + e.sum(1, keepdim=True).to(torch.int32)

    full = vd.sharded_call(fake_generate, [cond, extra], n_total)
    want = fake_generate(cond, extra)
    lo, hi = vd.shard_range(n_total, rank, world)
    q.put((rank, bool((full == want).all()), (lo, hi), tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_total, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_shard_ranges():
    sys.path.insert(0, ROOT)
    import video_llamagen_amd  # noqa: F401
    from video_llamagen_amd.dist import shard_range
    for n in (1, 7, 32, 33, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gather_even_and_ragged():
    for n in (8, 7):
        res = _run(n)
        assert [r[1] for r in res] == [True, True]
        assert res[0][2][1] == res[1][2][0] and res[1][2][1] == n
        assert res[0][3] == (n, 5)
