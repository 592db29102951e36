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


# ---- root-rank budget of the ONE collective ------------------------------------------------------------
# Measured on one MI355X with the root's inbound volume EMULATED (tools/gather_probe.py --beside --emulate-world 8: the N-1
# inbound shards as ncclSend/ncclRecv to self, so RCCL's SendRecv kernel really runs beside the forward; only the xGMI hop and
# the peers' clocks are missing): profiles/r04/gather_root_emulation_*.txt.  Findings the policy encodes:
#   * capping RCCL's channels makes the root SLOWER, not faster: 8 pairs per gather at 1 pair per step cost 5.1 % with RCCL's
#     default channel count, 5.9 / 6.6 / 8.5 / 11.2 % capped at 8 / 4 / 2 / 1 channels (the copy kernel then lives longer beside
#     the latency-bound chain) -> no cap;
#   * what the root pays is mostly the HOST time of the call (55-75 us for one grouped operation, 200-225 us for 14 Python-level
#     P2P operations) on a loop that is nearly host-bound at 1 pair per step, plus ~15 us of RCCL kernel per 29 MB;
#   * 8 pairs per GPU per step (BASELINE config 4): a gather every step costs 1.1 % (one grouped call) to 4.6 % (14 P2P
#     operations); every second step 0.8 % to 2.8 %.  1 pair per step: 3.3-5.1 % at 8 pairs per gather, 3.4-4.2 % at 16.
# Hence: 8 ranks gather 16 pairs per rank at a time (<= 3 % at 8 pairs per step, <= 5 % at 1 pair per step in the emulation's
# pessimistic variant), smaller worlds 8 pairs.  UNMEASURED ON N > 1: no multi-GPU box was available to this build.
GATHER_POLICY = {
    # world size (largest key <= world applies): (RCCL channel cap or None, minimum pairs per rank carried by one gather)
    1: (None, 8),
    2: (None, 8),
    4: (None, 8),
    8: (None, 16),
}


def gather_policy(world):
    """(channel cap or None, minimum pairs per rank carried by one gather) for a job of `world` ranks."""
    key = max(k for k in GATHER_POLICY if k <= max(1, int(world)))
    return GATHER_POLICY[key]


def gather_every(world, pairs_per_step, min_pairs=None):
    """Steps per gather for `pairs_per_step` pairs per rank per step: the smallest G with G * pairs_per_step >= the policy's
    minimum for this world size (or `min_pairs` when the caller overrides it)."""
    need = gather_policy(world)[1] if min_pairs is None else int(min_pairs)
    return max(1, -(-need // max(1, int(pairs_per_step))))


def apply_channel_cap(world):
    """Exports RCCL's channel caps for a `world`-rank job unless the caller set them; must run before the communicator exists.
    (The measured policy is "no cap" at every world size; the hook stays so that a node that measures otherwise changes a table.)"""
    cap, _ = gather_policy(world)
    if cap is not None:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(cap))
        os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", str(cap))
    return cap


def init_from_env(backend=None, tune=True, timeout_s=None):
    """Initialise the default process group from torchrun's environment.  Without that environment (plain
    `python bench.py`) this is a no-op; under torchrun a world of ONE is initialised too, so that a single GPU runs the
    same RCCL code path (communicator set-up, gather, barrier) the N-GPU job runs.  `tune`: apply gather_policy's channel
    cap for this world size first (tools/gather_probe.py passes False to measure RCCL's own defaults).
    `timeout_s`: the process group's timeout (rendezvous, communicator set-up and -- through torch's NCCL watchdog -- every
    collective): a rank that never arrives fails the others after this long instead of after torch's 10-minute default.
    On the RCCL backend a rank whose LOCAL_RANK has no device fails at once with a message that says so."""
    rank, local_rank, world = env_rank()
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if launched and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if timeout_s is not None:
            from datetime import timedelta
            kw["timeout"] = timedelta(seconds=float(timeout_s))
        if backend == "nccl":
            have = torch.cuda.device_count()                  # (does not initialise HIP)
            if local_rank >= have:
                raise RuntimeError(f"rank {rank}: LOCAL_RANK {local_rank} has no HIP device ({have} visible, the job has "
                                   f"{world} ranks): RCCL needs one device per rank")
            if tune:
                apply_channel_cap(world)
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def local_device(local_rank, share_one_gpu=False):
    """The HIP device of this rank: cuda:LOCAL_RANK, or cuda:0 for every rank when the ranks deliberately share one GPU
    (gloo only -- RCCL refuses two ranks on one device)."""
    if share_one_gpu:
        if dist.is_initialized() and dist.get_backend() != "gloo":
            raise RuntimeError("ranks can share one GPU on the gloo backend only (RCCL refuses duplicate devices)")
        return torch.device("cuda", 0)
    return torch.device("cuda", int(local_rank))


def shard_range(total, rank, world):
    """Contiguous, balanced split of `total` pairs: rank r takes [lo, hi)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _through_host(t, group=None):
    """True when the collective must carry a device tensor through host memory: the gloo backend has no device-side gather.
    That combination is the debugging / single-GPU form of the N-rank job (several ranks sharing one GPU: RCCL refuses
    duplicate devices in one communicator) -- tests/test_gpu_dist.py::test_two_ranks_on_one_gpu_*, `bench.py --one-gpu`."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _HostGather:
    """Work handle of a gloo gather of device tensors: the shard went to the host on the caller's stream (synchronised),
    gloo gathered host tensors asynchronously; wait() completes that and, on `dst`, copies the shards into the device buffers."""

    def __init__(self, work, host_bufs, bufs):
        self.work, self.host_bufs, self.bufs = work, host_bufs, bufs

    def wait(self):
        self.work.wait()
        if self.bufs is not None:
            for d, h in zip(self.bufs, self.host_bufs):
                d.copy_(h, non_blocking=False)
        return True


def gather_pairs(local, counts=None, dst=0, group=None):
    """Gathers per-rank disparity maps [b_r,1,H,W] to `dst` (concatenated in rank order); other ranks get None.

    Equal shards use one `gather`; ragged shards are padded to the largest shard first.  On the gloo backend device tensors
    travel through host memory (`_through_host`) and come back as a device tensor on `dst`."""
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
    device = send.device
    if _through_host(send, group):
        send = send.cpu()                                  # (synchronises the stream that produced the shard)
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0).to(device)


def gather_async(local, bufs, dst=0, group=None):
    """The ONE collective of the path as bench.py issues it per step: equal shards, buffers pre-allocated on `dst`
    (`bufs`: list of world tensors there, None elsewhere), asynchronous -- RCCL runs it on its own stream behind this
    step's kernels so it overlaps the next step.  Returns the work handle (wait() before reading `bufs`).
    On the gloo backend with device tensors (`_through_host`) the call is NOT asynchronous with respect to the device: the
    shard is copied to the host here, which blocks the calling thread until the stream that produced it has drained, and
    `wait()` copies the gathered shards back with blocking host-to-device copies.  That form exists to run the N-rank code
    where RCCL cannot (ranks sharing one GPU; the labelled fallback of bench.py); what it costs is not the collective's price."""
    # `dst` of dist.gather is a GLOBAL rank: compare with the global rank, not the group-local one
    root = dist.get_rank() == dst
    if _through_host(local, group):
        send = local.contiguous().cpu()
        host = [torch.empty_like(send) for _ in bufs] if root else None
        return _HostGather(dist.gather(send, host, dst=dst, group=group, async_op=True), host, bufs if root else None)
    return dist.gather(local.contiguous(), bufs if root else None, dst=dst, group=group, async_op=True)


class StagedGather:
    """The ONE collective of the path for a stream of small steps (bench.py: 1 pair per GPU per step): every rank writes
    the stage-4 maps of `group` consecutive steps into the slots of a staging buffer [group*B,1,H,W] -- `slot()` is the
    destination handed to the forward, so no copy is made -- and `commit()` after each step issues ONE asynchronous
    gather to rank `dst` when the buffer is full.  Two staging buffers alternate: the gather of one overlaps the steps that
    fill the other.  `flush()` gathers a partly filled buffer (the tail) and waits for everything in flight.
    Rank `dst` reads the gathered maps of rank r, gather g (0 = most recent completed) from `gathered(r)`.

    With `multi_stream=True` steps may run on DIFFERENT streams (bench.py --streams N): `commit()` notes the stream that
    produced the slot and the gather is ordered behind every such stream (one event per stream per gather); `slot()` makes a
    stream wait, once, until the previous gather out of the buffer it hands out has completed (the gather used
    to be ordered behind the last committing stream only).
    On the gloo backend with device buffers `commit()` blocks the host whenever it issues a gather (see gather_async)."""

    def __init__(self, B, H, W, group, device, dtype=torch.float32, dst=0, multi_stream=False):
        self.B, self.group, self.dst = int(B), max(1, int(group)), dst
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        shape = (self.group * self.B, 1, H, W)
        self.staging = [torch.empty(shape, device=device, dtype=dtype) for _ in range(2)]
        self.recv = [[torch.empty(shape, device=device, dtype=dtype) for _ in range(self.world)] for _ in range(2)] \
            if self.rank == dst else [None, None]
        self.pending = [None, None]
        self.buf, self.fill, self.count = 0, 0, 0
        self.last = None                      # (buffer, slots filled) of the most recent gather
        self.on_gpu = bool(multi_stream) and torch.device(device).type == "cuda"   # events only when steps use several streams
        self.slot_streams = [set(), set()]    # per staging buffer: the streams that committed slots into it
        self.free_event = [None, None]        # per staging buffer: (event recorded once its previous gather completed, streams that waited)

    def slot(self):
        """Destination [B,1,H,W] for this step's stage-4 map (valid on the CURRENT stream)."""
        if self.on_gpu:
            fe = self.free_event[self.buf]
            if fe is not None:                               # once per (buffer release, stream), not once per step
                ev, seen = fe
                cur = torch.cuda.current_stream()
                if cur not in seen:
                    cur.wait_event(ev)
                    seen.add(cur)
        return self.staging[self.buf][self.fill * self.B:(self.fill + 1) * self.B]

    def _issue(self):
        b = self.buf
        if self.on_gpu:
            # order the gather behind every stream that wrote a slot of this buffer: one event per such stream, recorded now
            # (it covers everything queued on that stream so far, i.e. at least its slots)
            cur = torch.cuda.current_stream()
            for st in self.slot_streams[b]:
                if st != cur:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    cur.wait_event(ev)
            self.slot_streams[b] = set()
        if dist.is_initialized():
            self.pending[b] = gather_async(self.staging[b], self.recv[b], dst=self.dst)
        else:
            self.recv[b][0].copy_(self.staging[b])
        self.last = (b, self.fill)
        self.count += 1
        self.buf, self.fill = 1 - b, 0
        if self.pending[1 - b] is not None:   # the buffer the next steps write into: its gather must have finished
            self.pending[1 - b].wait()        # (a stream-side wait on the NCCL backend, not a host block)
            self.pending[1 - b] = None
            if self.on_gpu:                   # ... and steps on other streams learn about it through this event
                ev = torch.cuda.Event()
                ev.record()
                self.free_event[1 - b] = (ev, {torch.cuda.current_stream()})

    def commit(self):
        """Call after the forward that wrote slot(), on the same stream; returns True when this step triggered a gather."""
        if self.on_gpu:
            self.slot_streams[self.buf].add(torch.cuda.current_stream())
        self.fill += 1
        if self.fill == self.group:
            self._issue()
            return True
        return False

    def warm(self):
        """One gather out of each staging buffer, waited for, BEFORE anything is timed: the first collective on a
        communicator pays its one-time set-up (channels, proxy threads, kernel load: ~5 ms on a world of one) and with G steps
        per gather a short warm-up never reaches the first gather -- round 4 measured 25 % "overhead" over 40 steps from
        exactly that.  Leaves the object as constructed."""
        if dist.is_initialized():
            for b in (0, 1):
                gather_async(self.staging[b], self.recv[b], dst=self.dst).wait()
            if self.staging[0].is_cuda:
                torch.cuda.synchronize(self.staging[0].device)
        self.pending = [None, None]
        self.buf, self.fill, self.count, self.last = 0, 0, 0, None

    def flush(self):
        """Gathers a partly filled buffer and waits (stream-side on NCCL) for everything in flight.  Both staging buffers are
        free afterwards; with `multi_stream` steps on other streams learn that through `free_event`, exactly as after
        `_issue` (flush() used to order the current stream only, so a step on another stream could overwrite a
        slot its tail gather was still reading)."""
        if self.fill > 0:
            self._issue()
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None
                if self.on_gpu:
                    ev = torch.cuda.Event()
                    ev.record()
                    self.free_event[i] = (ev, {torch.cuda.current_stream()})

    def reset(self):
        """After flush(): start filling at slot 0 of buffer 0 again and zero the gather count (bench.py between the warm-up
        and the timed region).  Pending free events stay in force."""
        if self.fill or any(w is not None for w in self.pending):
            raise RuntimeError("StagedGather.reset() needs a flush() first")
        self.buf, self.fill, self.count = 0, 0, 0

    def gathered(self, rank):
        """On `dst` after flush(): (tensor [group*B,1,H,W] of `rank`'s most recent gather, number of valid steps in it)."""
        b, n = self.last
        return self.recv[b][rank], n


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
