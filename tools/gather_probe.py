#!/usr/bin/env python3
"""How long does ONE RCCL gather of the stage-4 maps take on an otherwise idle GPU (world of one under torchrun, or N ranks)?
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29540 tools/gather_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import torch.distributed as dist
from lwsnet_amd import dist as ldist
rank, local, world = ldist.init_from_env()
dev = torch.device("cuda", local)
for pairs in (1, 2, 4, 8, 16, 32):
    x = torch.randn((pairs, 1, 256, 512), device=dev)
    bufs = [torch.empty_like(x) for _ in range(world)] if rank == 0 else None
    for _ in range(3):
        ldist.gather_async(x, bufs).wait()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        w = ldist.gather_async(x, bufs)
        t1 = time.perf_counter()
        w.wait()
        torch.cuda.synchronize()
        ts.append((t1 - t0, time.perf_counter() - t0))
    ts.sort(key=lambda v: v[1])
    if rank == 0:
        print(f"{pairs:3d} pairs = {x.numel() * 4 / 1e6:6.2f} MB per rank: host call {1e6 * ts[5][0]:7.1f} us, issue -> complete {1e6 * ts[5][1]:8.1f} us (median of 10, world {world})", flush=True)
dist.destroy_process_group()
