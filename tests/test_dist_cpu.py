"""N > 1 plumbing on CPU: world_size-2 gloo processes shard a batch of pairs, run a per-pair stand-in for the
device forward, and gather the stage-4 maps on rank 0 -- the result must equal the unsharded run bit for bit
(pure partitioning)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lwsnet_amd.dist import gather_pairs, shard_range, sharded_forward


def test_shard_range_partitions():
    for total in (1, 2, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_forward(left, right):
    """Per-sample function of the pair (no cross-sample op), like the eval-mode network."""
    d = (left - right).abs().sum(1, keepdim=True)
    return [d * (s + 1) + torch.arange(d.shape[-1], dtype=d.dtype) for s in range(4)]


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        left = torch.randn(B, 3, 8, 16, generator=g)
        right = torch.randn(B, 3, 8, 16, generator=g)
        preds, gathered = sharded_forward(_fake_forward, left, right)
        lo, hi = shard_range(B, rank, world)
        assert preds[3].shape[0] == hi - lo
        if rank == 0:
            q.put(gathered.numpy())
        else:
            assert gathered is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_sharded_gather_equals_unsharded(B):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    left = torch.randn(B, 3, 8, 16, generator=g)
    right = torch.randn(B, 3, 8, 16, generator=g)
    want = _fake_forward(left, right)[3].numpy()
    assert np.array_equal(got, want)


def test_gather_is_identity_without_process_group():
    x = torch.ones(2, 1, 4, 4)
    assert gather_pairs(x) is x


def test_staged_gather_flush_then_reset():
    """StagedGather.reset() (callers used to poke buf / fill): refuses while slots are filled, and after flush()
    starts at slot 0 of buffer 0 with a zero gather count.  No process group: the 'gather' is a copy on this rank."""
    from lwsnet_amd.dist import StagedGather
    sg = StagedGather(1, 4, 8, 3, torch.device("cpu"))
    for k in range(4):
        sg.slot().fill_(float(k))
        sg.commit()
    assert sg.count == 1 and sg.fill == 1
    with pytest.raises(RuntimeError, match="flush"):
        sg.reset()
    sg.flush()
    got, nvalid = sg.gathered(0)
    assert sg.count == 2 and nvalid == 1 and float(got[0, 0, 0, 0]) == 3.0
    sg.reset()
    assert (sg.buf, sg.fill, sg.count) == (0, 0, 0)


def test_package_import_exports_the_runtime_switches():
    """lwsnet_amd/__init__.py: GPU_MAX_HW_QUEUES and HSA_ENABLE_IPC_MODE_LEGACY are exported by ANY import of the package
    (bench.py used to be the only entry point that set them), the caller's values win, and an import after
    HIP is up is reported by late_env()."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import os, sys; sys.path.insert(0, %r); import lwsnet_amd.dist, lwsnet_amd; "
            "print(os.environ['GPU_MAX_HW_QUEUES'], os.environ['HSA_ENABLE_IPC_MODE_LEGACY'], lwsnet_amd.late_env())" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and out.stdout.split()[:2] == ["8", "0"] and out.stdout.strip().endswith("[]"), (out.stdout, out.stderr)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES="4"), timeout=300)
    assert out.stdout.split()[:2] == ["4", "0"], (out.stdout, out.stderr)
