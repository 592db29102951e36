#!/usr/bin/env python3
"""What the ONE collective of the path (gather of stage-4 maps to rank 0) costs on the ROOT rank.

Two modes, both under a real RCCL communicator (world of one under torchrun, or N ranks on an N-GPU node):

  idle    (default)  one gather of 0.5 ... 16.8 MB on an otherwise idle GPU: host call and issue -> complete time
  --beside           K forwards with an asynchronous gather every G steps against the same K forwards without it.
                     With --emulate-world N (world of one) the message is N x the rank's own G*B maps, i.e. the bytes the
                     root of an N-rank job writes per gather (N-1 inbound shards + its own): RCCL's send/recv kernel then
                     moves the root's whole inbound volume on this GPU's CUs and memory system; only the xGMI hop and the
                     peers' clocks are missing.  Channel caps are read by RCCL at communicator creation, so they are set in
                     the environment of the process (NCCL_MAX_NCHANNELS, NCCL_MAX_P2P_NCHANNELS, ...); the line printed
                     records the NCCL_* / RCCL_* variables it ran with.

    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29540 tools/gather_probe.py
    NCCL_MAX_NCHANNELS=2 NCCL_MAX_P2P_NCHANNELS=2 python -m torch.distributed.run ... tools/gather_probe.py --beside \
        --batch 1 --emulate-world 8 --gather-pairs 8,16,32,64
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402
from lwsnet_amd import dist as ldist  # noqa: E402


def idle(rank, world, dev):
    for pairs in (1, 2, 4, 8, 16, 32, 64):
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
            print(f"{pairs:3d} pairs = {x.numel() * 4 / 1e6:6.2f} MB per rank: host call {1e6 * ts[5][0]:7.1f} us, issue -> complete "
                  f"{1e6 * ts[5][1]:8.1f} us (median of 10, world {world})", flush=True)


def beside(a, rank, world, dev):
    from lwsnet_amd import _lib
    from lwsnet_amd.models import LWSNet
    from lwsnet_amd.synth import make_batch
    from lwsnet_amd.weights import default_args, make_state_dict
    H, W = [int(v) for v in a.size.split("x")]
    B = a.batch
    model = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
    lib = _lib.load()
    _lib.check(lib.lws_reserve(model._h, B, H, W), "lws_reserve")
    ln, rn = make_batch(B, H, W, first_index=rank * B)
    left, right = torch.from_numpy(ln).to(dev), torch.from_numpy(rn).to(dev)
    env = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "GPU_MAX_HW"))}
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:                       # clock ramp of an idle GPU
        for _ in range(25):
            model(left, right)
        torch.cuda.synchronize()

    def issue_gather(stage_b, recv_b, mult):
        """The root's side of one gather.  world > 1: the product's own call (lwsnet_amd.dist.gather_async).  World of one:
        torch's NCCL gather copies the root's own shard with a tensor copy and posts ncclRecv for the OTHER ranks only, so a
        world-of-one gather never launches an RCCL kernel (r03's world-of-one figures measured that copy).  The emulation
        therefore posts, in ONE ncclGroup, mult-1 ncclSend/ncclRecv pairs to SELF of one shard each -- RCCL's SendRecv
        kernel then moves the N-1 inbound shards through its channels on this GPU -- plus the own-shard copy."""
        if world > 1 or mult == 1 or a.self_copy:
            return ldist.gather_async(stage_b, recv_b)
        n = stage_b.shape[0] // mult
        ops = []
        if a.one_op:                  # the same bytes as ONE send/recv pair (host cost of one group with two operations)
            ops = [dist.P2POp(dist.isend, stage_b[n:], 0), dist.P2POp(dist.irecv, recv_b[0][n:], 0)]
        else:
            for i in range(1, mult):
                ops.append(dist.P2POp(dist.isend, stage_b[i * n:(i + 1) * n], 0))
                ops.append(dist.P2POp(dist.irecv, recv_b[0][i * n:(i + 1) * n], 0))
        works = dist.batch_isend_irecv(ops)
        recv_b[0][:n].copy_(stage_b[:n], non_blocking=True)

        class _W:
            def wait(self_inner):
                for w in works:
                    w.wait()
        return _W()

    host_us = []

    def run(steps, G, mult, issue=True):
        """K steps; G > 0: the stage-4 maps of G consecutive steps land in the slots of a staging buffer (the forward's
        output pointer) and ONE asynchronous gather of `mult` x that buffer follows; two buffers alternate.
        issue=False: the slots are written but no gather is issued (what the staging itself costs)."""
        stage = [torch.empty((mult * max(G, 1) * B, 1, H, W), device=dev) for _ in range(2)]
        recv = [[torch.empty_like(stage[0]) for _ in range(world)] if rank == 0 else None for _ in range(2)]
        pend = [None, None]
        host_us.clear()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b, fill = 0, 0
        for _ in range(steps):
            if G > 0:
                model(left, right, out=[None, None, None, stage[b][fill * B:(fill + 1) * B]])
                fill += 1
                if fill == G:
                    if issue:
                        th = time.perf_counter()
                        pend[b] = issue_gather(stage[b], recv[b], mult)
                        host_us.append(1e6 * (time.perf_counter() - th))
                    b, fill = 1 - b, 0
                    if pend[b] is not None:
                        pend[b].wait()
                        pend[b] = None
            else:
                model(left, right)
        for w in pend:
            if w is not None:
                w.wait()
        torch.cuda.synchronize()
        dist.barrier()
        return (time.perf_counter() - t0) / steps

    mult = max(1, a.emulate_world) if world == 1 else 1
    for gp in [int(v) for v in a.gather_pairs.split(",")]:
        G = max(1, -(-gp // B))
        steps = max(a.steps, 6 * G)
        steps -= steps % G
        run(2 * G, G, mult)
        plain, slots, with_g = [], [], []
        for _ in range(a.reps):                              # interleaved, so that clock drift hits all alike
            plain.append(run(steps, 0, 1))
            slots.append(run(steps, G, mult, issue=False))
            with_g.append(run(steps, G, mult))
            host_issue = sorted(host_us)[len(host_us) // 2] if host_us else None
        med = lambda v: sorted(v)[len(v) // 2]               # noqa: E731
        p, sl, g = med(plain), med(slots), med(with_g)
        if rank == 0:
            print(json.dumps({"batch": B, "size": a.size, "world": world, "emulated_world": mult if world == 1 else None,
                              "emulation": None if world > 1 else ("torch gather (tensor copy, no RCCL kernel)" if (a.self_copy or mult == 1)
                                                                   else (f"{'1 ncclSend/ncclRecv pair' if a.one_op else str(mult - 1) + ' ncclSend/ncclRecv pairs'} to self in one group + own-shard copy")),
                              "pairs_per_rank_per_gather": G * B, "gather_every_steps": G,
                              "MB_written_on_root_per_gather": round(mult * world * G * B * H * W * 4 / 1e6, 2) if world == 1 else
                              round(world * G * B * H * W * 4 / 1e6, 2),
                              "ms_per_step_plain": round(1e3 * p, 4), "ms_per_step_slots_only": round(1e3 * sl, 4),
                              "ms_per_step_with_gather": round(1e3 * g, 4),
                              "overhead_pct": round(100.0 * (g - p) / p, 2), "overhead_pct_vs_slots_only": round(100.0 * (g - sl) / sl, 2),
                              "overhead_pct_min_max": [round(100.0 * (min(with_g) - max(plain)) / max(plain), 2),
                                                       round(100.0 * (max(with_g) - min(plain)) / min(plain), 2)],
                              "host_us_per_gather_call": None if host_issue is None else round(host_issue, 1),
                              "steps": steps, "reps": a.reps, "env": env}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--beside", action="store_true")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", default="256x512")
    ap.add_argument("--emulate-world", type=int, default=8)
    ap.add_argument("--gather-pairs", default="8")
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--one-op", action="store_true", help="emulation with ONE send/recv pair to self carrying all inbound shards")
    ap.add_argument("--self-copy", action="store_true", help="world of one: torch's own gather (a tensor copy) instead of ncclSend/ncclRecv to self")
    a = ap.parse_args()
    rank, local, world = ldist.init_from_env(tune=False)
    if not dist.is_initialized():
        raise SystemExit("run under torch.distributed.run (a world of one is fine)")
    dev = torch.device("cuda", local)
    if a.beside:
        beside(a, rank, world, dev)
    else:
        idle(rank, world, dev)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
