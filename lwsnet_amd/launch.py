"""Job control of the N-rank bench (BASELINE config 4): one SUPERVISOR per rank that never touches the GPU.

The reference has no multi-process path at all (/root/reference/inference.py:38 is one device, one process; pairs are
independent, models/models.py:106-164), so nothing here mirrors reference code: this is the harness that makes the first
contact of the batch-sharded job with RCCL impossible to waste.

Whoever starts `bench.py --gpus N` under `torch.distributed.run` (the driver does, `bench.py` itself does for a bare
`python bench.py --gpus N`) gets N supervisor processes.  A supervisor

* joins a gloo control group with the other supervisors (CPU only -- it never initialises HIP, so it may start children
  freely; nothing is ever exec'ed);
* local rank 0 builds the HIP library ONCE for the node, before any worker loads it;
* starts its rank's WORKER as a fresh child in its own process group (`LWS_BENCH_WORKER=1`, a rendezvous port of its own) and
  watches it: a deadline (`--job-timeout`), the worker's exit code, and a failure counter in the control group's store so that
  one failed rank ends every other rank's attempt at once instead of after a collective timeout;
* if the RCCL attempt failed anywhere, ALL supervisors start ONE fallback attempt in fresh workers -- the same job with the
  stage-4 gather carried over gloo through host memory (the path tests/test_gpu_dist.py proves on one GPU) -- and rank 0 labels
  the line `collective.backend: "gloo-through-host (RCCL job failed: ...)"`;
* rank 0 relays exactly ONE JSON line: the worker's, or `{"value": null, "error": ...}` with a non-zero exit status.
"""
from __future__ import annotations

import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time
from datetime import timedelta

METRIC = "stereo pairs/sec @256x512 maxdisp=192 (stage-4)"
ATTEMPTS = (("rccl", []), ("gloo-host", ["--collective", "gloo-host"]))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def error_line(n_gpus, steps, warmup, error, **more):
    """The one JSON line of a job that measured nothing."""
    d = {"metric": METRIC, "value": None, "unit": "pairs/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
         "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic", "error": error}
    d.update(more)
    return json.dumps(d)


def kill_group(p, grace=5.0):
    """Ends the process group of a child started with start_new_session=True (the exact group we created)."""
    if p.poll() is not None:
        return
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(p.pid, sig)
        except ProcessLookupError:
            return
        t0 = time.monotonic()
        while time.monotonic() - t0 < grace:
            if p.poll() is not None:
                return
            time.sleep(0.05)


class Child:
    """A worker process in its own process group; stdout is collected line by line, stderr is inherited."""

    def __init__(self, cmd, env, cwd=None):
        self.p = subprocess.Popen(cmd, env=env, cwd=cwd, stdout=subprocess.PIPE, text=True, start_new_session=True)
        self.lines = []
        self._t = threading.Thread(target=self._read, daemon=True)
        self._t.start()

    def _read(self):
        for line in self.p.stdout:
            self.lines.append(line.rstrip("\n"))

    def watch(self, deadline, peer_failed=lambda: False, poll=0.2):
        """Waits for the child.  Returns (ok, reason): the deadline and a failure of any peer both end the child's group."""
        while True:
            rc = self.p.poll()
            if rc is not None:
                self._t.join(timeout=5.0)
                return (rc == 0), (None if rc == 0 else f"worker exited with status {rc}")
            if time.monotonic() > deadline:
                kill_group(self.p)
                self._t.join(timeout=5.0)
                return False, "worker hit the job timeout and was killed"
            if peer_failed():
                kill_group(self.p)
                self._t.join(timeout=5.0)
                return False, "ended because another rank failed"
            time.sleep(poll)

    def json_lines(self):
        return [l for l in self.lines if l.startswith("{")]


def worker_env(base, rank, local_rank, world, port, attempt):
    """Environment of a worker: torchrun's rank variables, a rendezvous of its own (the agent's store on MASTER_PORT belongs to
    the supervisors: TORCHELASTIC_USE_AGENT_STORE must not reach the worker, rank 0 of the workers hosts their store)."""
    env = {k: v for k, v in base.items() if not k.startswith("TORCHELASTIC_")}
    env.update({"RANK": str(rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(port), "LWS_BENCH_WORKER": "1", "LWS_BENCH_ATTEMPT": str(attempt)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def _first_reason(store, attempt, world):
    """The first failure message any supervisor left in the control store for this attempt."""
    for r in range(world):
        key = f"lws/a{attempt}/reason/{r}"
        try:
            if store.check([key]):
                msg = store.get(key).decode("utf-8", "replace")
                return msg if msg.startswith("rank ") else f"rank {r}: {msg}"
        except Exception:
            pass
    return "unknown"


def supervise(script, argv, args, visible_gpus=None):
    """Body of a supervisor (one per torchrun rank).  `script`/`argv`: the worker command is
    `python script argv [attempt-specific flags]`.  Returns the exit status."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ["WORLD_SIZE"])
    t_start = time.monotonic()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=max(120, int(2.5 * args.job_timeout))))
    store = dist.distributed_c10d._get_default_store()
    cpu_only = bool(args.dry_run_cpu)
    shared = bool(getattr(args, "one_gpu", False))

    def say(line):
        if rank == 0:
            print(line, flush=True)

    def agree(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    # (1) enough devices?  (device_count does not initialise HIP)  A rank without a device cannot be helped by a fallback.
    if not cpu_only:
        have = torch.cuda.device_count() if visible_gpus is None else visible_gpus
        need = 1 if shared else world
        if not agree(have >= need):
            say(error_line(world, args.steps, args.warmup,
                           f"{have} HIP device(s) visible on this node, the job needs {need} (one per rank): nothing was run"))
            dist.destroy_process_group()
            return 4
    # (2) ONE build for the node, before any worker loads the library
    built = True
    if local_rank == 0 and not cpu_only:
        try:
            from . import build
            build.build_library()
        except Exception as e:                                   # noqa: BLE001 (reported, every rank leaves together)
            built = False
            sys.stderr.write(f"[bench supervisor {rank}] building liblwsnet_hip.so failed: {e}\n")
    if not agree(built):
        say(error_line(world, args.steps, args.warmup, "building liblwsnet_hip.so failed (see stderr): nothing was run"))
        dist.destroy_process_group()
        return 5
    if shared or args.collective == "gloo-host":
        attempts = (("gloo-host", []),)                          # (the flag is already in argv; nothing to fall back to)
    elif cpu_only:
        attempts = (("gloo", []), ATTEMPTS[1])                   # the dry run exercises the same two-attempt protocol
    else:
        attempts = ATTEMPTS
    log = []
    status, line = 1, None
    for k, (name, extra) in enumerate(attempts):
        port = torch.tensor([free_port() if rank == 0 else 0], dtype=torch.int64)
        dist.broadcast(port, src=0)
        env = worker_env(os.environ, rank, local_rank, world, int(port.item()), k)
        child = Child([sys.executable, script, *argv, *extra], env)
        fail_key = f"lws/a{k}/failed"
        ok, reason = child.watch(time.monotonic() + args.job_timeout, lambda: store.add(fail_key, 0) > 0)
        if not ok:
            store.add(fail_key, 1)
            if reason != "ended because another rank failed":
                errs = [json.loads(l).get("error") for l in child.json_lines() if '"error"' in l]
                store.set(f"lws/a{k}/reason/{rank}", (errs[-1] if errs and errs[-1] else reason)[:400])
        for l in child.lines:                                   # whatever else a worker printed goes to stderr
            if not l.startswith("{"):
                sys.stderr.write(f"[worker {rank}] {l}\n")
        all_ok = agree(ok)
        if all_ok:
            if rank == 0:
                js = [l for l in child.json_lines() if '"error"' not in l]
                if len(js) == 1:
                    line = json.loads(js[0])
                else:
                    all_ok = False
                    store.set(f"lws/a{k}/reason/0", f"rank 0 printed {len(js)} JSON lines instead of one")
            all_ok = agree(all_ok)
        why = None if all_ok else _first_reason(store, k, world)
        log.append({"collective": name, "ok": all_ok, **({} if all_ok else {"reason": why}), "s": round(time.monotonic() - t_start, 1)})
        if all_ok:
            status = 0
            break
    if rank == 0:
        if status == 0:
            if len(log) > 1 and "collective" in line:
                first = {"rccl": "RCCL"}.get(log[0]["collective"], log[0]["collective"])
                line["collective"]["backend"] = f"gloo-through-host ({first} job failed: {log[0].get('reason')})"
            line["attempts"] = log
            print(json.dumps(line), flush=True)
        else:
            print(error_line(world, args.steps, args.warmup, "; ".join(f"{a['collective']}: {a.get('reason')}" for a in log),
                             attempts=log), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return status


def self_launch(script, argv, n, job_timeout, steps, warmup):
    """`python bench.py --gpus N` outside torchrun: start `python -m torch.distributed.run ... bench.py` as a CHILD in its own
    process group (this parent never touches the GPU and never execs), with a deadline that covers both attempts of the
    supervisors plus start-up; relays the child's output."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    child = Child(cmd, env)
    ok, reason = child.watch(time.monotonic() + 2.0 * job_timeout + 180.0)
    js = child.json_lines()
    for l in child.lines:
        if not l.startswith("{"):
            sys.stderr.write(l + "\n")
    if js:
        print(js[-1], flush=True)
        return 0 if ok else (child.p.returncode or 1)
    print(error_line(n, steps, warmup, f"the {n}-rank job printed no JSON line ({reason})"), flush=True)
    return child.p.returncode or 1


def arm_watchdog(seconds, on_expire):
    """In-process deadline for a worker: a daemon thread that reports and leaves with os._exit (the main thread may be blocked
    inside a HIP / RCCL call that never returns)."""
    def run():
        time.sleep(seconds)
        try:
            on_expire()
        finally:
            os._exit(3)
    t = threading.Thread(target=run, daemon=True)
    t.start()
    return t
