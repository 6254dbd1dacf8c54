"""Batch sharding + the one final gather.

Every sample is independent (no cross-sample op in generate or the decoders; CFG pairs stay on one GPU), so ranks take
contiguous batch shards and never communicate inside the path.  The only exchange is the final gather of token ids /
latents / frames: one `all_gather` (RCCL over xGMI on the GPU box, gloo in the CPU tests).
In-repo analogue in the reference: tokenizer/tokenizer_image/reconstruction_vq_ddp.py:100-115,156-161.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """env:// rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*), one process per GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank`; the first n % world ranks get one extra item."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_batch(local, n_total):
    """Concatenates the ranks' shards (dim 0) in rank order on every rank; shards may differ by one row."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    per = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad) if local.is_cuda else dist.all_gather(list(out.view((world, per) + tuple(local.shape[1:])).unbind(0)), pad)
    out = out.view((world, per) + tuple(local.shape[1:]))
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r, : hi - lo])
    return torch.cat(parts, 0)


def sharded_call(fn, batch_inputs, n_total):
    """Runs fn on this rank's contiguous shard of every tensor in `batch_inputs` and gathers the result."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(n_total, rank, world)
    local = fn(*[None if t is None else t[lo:hi] for t in batch_inputs])
    return gather_batch(local, n_total)
