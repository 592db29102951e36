"""Batch sharding over the GPUs of one node.

Stereo pairs are independent in eval mode (BatchNorm uses running statistics and every op of
/root/reference/models/models.py:106-164 is per sample), so the path shards by pairs with no
data-path collective; the only exchange is ONE gather of the stage-4 maps to the root rank.
One process per GPU; `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm, "gloo" on
CPU (used by the world_size-2 tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment.  Without that environment (plain
    `python bench.py`) this is a no-op; under torchrun a world of ONE is initialised too, so that a single GPU runs the
    same RCCL code path (communicator set-up, gather, barrier) the N-GPU job runs."""
    rank, local_rank, world = env_rank()
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if launched and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(total, rank, world):
    """Contiguous, balanced split of `total` pairs: rank r takes [lo, hi)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_pairs(local, counts=None, dst=0, group=None):
    """Gathers per-rank disparity maps [b_r,1,H,W] to `dst` (concatenated in rank order); other ranks get None.

    Equal shards use one `gather`; ragged shards are padded to the largest shard first."""
    if not dist.is_initialized():
        return local
    world, rank = dist.get_world_size(group), dist.get_rank()        # (global rank: `dst` is one)
    if counts is None:
        counts = [local.shape[0]] * world
    bmax = max(counts)
    send = local
    if local.shape[0] != bmax:
        pad = torch.zeros((bmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], 0)
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


def gather_async(local, bufs, dst=0, group=None):
    """The ONE collective of the path as bench.py issues it per step: equal shards, buffers pre-allocated on `dst`
    (`bufs`: list of world tensors there, None elsewhere), asynchronous -- RCCL runs it on its own stream behind this
    step's kernels so it overlaps the next step.  Returns the work handle (wait() before reading `bufs`)."""
    # `dst` of dist.gather is a GLOBAL rank: compare with the global rank, not the group-local one
    return dist.gather(local.contiguous(), bufs if dist.get_rank() == dst else None, dst=dst, group=group, async_op=True)


def sharded_forward(model_fn, left, right, dst=0):
    """Runs `model_fn(left_shard, right_shard) -> [4 x [b,1,H,W]]` on this rank's shard of the global batch
    and gathers the stage-4 maps on `dst`.  `left`/`right` are the GLOBAL batch (every rank holds or can
    index it); returns (local_preds, gathered_stage4_or_None)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    B = left.shape[0]
    if B < world:
        # some rank would get an empty shard: lws_forward needs B >= 1, and a rank that raises while the others sit
        # in the gather hangs the job -- so every rank raises the same error up front
        raise ValueError(f"global batch {B} is smaller than the world size {world}: every rank needs at least one pair")
    lo, hi = shard_range(B, rank, world)
    preds = model_fn(left[lo:hi], right[lo:hi])
    counts = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    return preds, gather_pairs(preds[3], counts, dst)
