#!/usr/bin/env python3
"""bench.py -- stereo pairs/s through LWSNet.forward at 256x512, maxdisplist=[24,5,5] (BASELINE.json).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one forward over this rank's batch of synthetic pairs (inputs and weights resident in HBM),
returning the stage-4 disparity; for N > 1 every rank runs its own pairs (weak scaling, no data-path
collective) and the stage-4 maps are gathered on rank 0 with ONE RCCL gather per step.
Rank 0 prints one JSON line.  `roofline` describes the dominant kernel (stage-1 Conv3D 32->32 on fp32
MFMA): algorithmic FLOPs per launch / average launch duration from hipEvents recorded by the library on
the launch stream inside the timed region; `roofline.traffic` = 2 x FETCH_SIZE + WRITE_SIZE of that kernel from two `rocprofv3
--pmc` child passes of the same workload that a plain single-GPU run starts before it touches the GPU (separate passes, no trace
domain; the committed summary under profiles/ is reported when the profiler is not available).  `cpu_baseline` is the literal oracle (torch-CPU ops) timed on
the host cores of the same box, N = 1 only; it also carries the numerics account: per-stage max-abs distance of the GPU
result and of the float32 literal oracle to the float64 literal oracle, for the smooth pair and for a white-noise pair.

Job control (lwsnet_amd/launch.py).  Under torchrun every rank process is a SUPERVISOR that never touches the GPU: local rank
0 builds the library once, each supervisor starts its rank's worker as a fresh child in its own process group and watches it
(`--job-timeout`, exit status, a failure flag shared over a gloo control group), and if the RCCL job fails or hangs anywhere the
supervisors start ONE fallback job in fresh workers whose gather goes over gloo through host memory, labelled as such in
`collective.backend`.  Rank 0 always prints exactly one JSON line -- with `"value": null, "error": ...` and a non-zero exit status
if nothing could be measured.  `python bench.py --gpus N` without torchrun's environment starts that torchrun job itself as a
child (never an exec) and relays the line.  Workers pass `timeout=120 s` to init_process_group (torch's NCCL watchdog applies it
to every collective) and fail at once when their LOCAL_RANK has no device.
With N > 1 (or --config4) the same job also times BASELINE config 4's per-rank shape -- 8 pairs per GPU per step, batch 8 N in
all -- and reports it under `config4`; `value` stays the 1-pair-per-GPU number so that N = 1 agrees with the single-GPU line.
`--dry-run-cpu` runs the same launch / supervise / shard / gather / report plumbing on CPU over gloo with a per-pair stand-in
for the forward (no GPU, nothing measured: "value" is null) -- it exists for tests/test_bench_cpu.py, which also injects a hung
rank and an init failure (LWS_BENCH_INJECT) to see the watchdog and the labelled fallback.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin; two streams on one queue serialise.
# A forward uses 2 streams (the caller's + the handle-owned side stream); PyTorch, RCCL (under torchrun) and the lws_pool
# workers add theirs.  Measured r03 on one MI355X: under `torchrun --nproc-per-node 1` the default of 4 cost 12 % of every
# step (0.563 vs 0.505 ms, with or without the gather) and the 4-worker pool 13 % (2,390 vs 2,740 pairs/s); 8 queues remove
# both and leave the plain single-stream run unchanged (1,972 vs 1,975 pairs/s).  Read once, when HIP initialises.
# Read once, when HIP initialises -- so lwsnet_amd/__init__.py exports it (and HSA_ENABLE_IPC_MODE_LEGACY=0, which RCCL needs on
# this pool) for EVERY entry point that imports the package before touching the GPU: a driver that starts
# `torchrun ... bench.py --gpus 8` directly gets them on every rank, as do lwsnet_amd.inference and users of model.pool().
import lwsnet_amd    # noqa: E402,F401  (before torch initialises HIP)

import numpy as np   # noqa: E402
import torch         # noqa: E402

H, W = 256, 512
PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2516.6        # same table: 16 x the f32-input rate ("~2.5 PF dense")


def _cpu_quota():
    """The CPU quota the box gives this job (cgroup v2 cpu.max / v1 cfs quota), as the file says it; '?' when unreadable."""
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(f) as fh:
                return fh.read().strip()
        except OSError:
            pass
    return "?"


def _sha256(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def _with_traffic(roof, B, H, W, maxdisp0, fp16):
    """roofline.traffic: HBM-side bytes per launch of the dominant kernel from the newest committed PMC summary OF THIS
    WORKLOAD (profiles/r*/pmc_fetch_write_b<B>_<H>x<W>.json, written by tools/pmc_summary.py from separate FETCH_SIZE /
    WRITE_SIZE passes, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read from inside this
    process, so the number is only reported when (a) a summary exists for exactly this batch, size and maxdisplist and (b) it
    records the sha256 of lwsnet_amd/csrc/lws_conv3d.hip it was collected with and that still matches the source in this
    tree; otherwise traffic stays null and traffic_note says why."""
    if roof is None:
        return roof
    if maxdisp0 != 24 or fp16:
        roof["traffic_note"] = "null: no PMC summary was collected for this maxdisplist / feature precision"
        return roof
    import glob
    tag = f"b{B}_{H}x{W}"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_fetch_write_{tag}.json")))
    if not files:
        roof["traffic_note"] = f"null: no PMC summary for this workload under profiles/ (pmc_fetch_write_{tag}.json)"
        return roof
    try:
        with open(files[-1]) as f:
            summary = json.load(f)
        want = (summary.get("kernel_source_sha256") or {}).get("lws_conv3d.hip")
        have = _sha256(os.path.join(ROOT, "lwsnet_amd", "csrc", "lws_conv3d.hip"))
        if want != have:
            roof["traffic_note"] = (f"null: {os.path.relpath(files[-1], ROOT)} was collected with a different lws_conv3d.hip "
                                    "(re-run tools/profile_run.sh + tools/collect_profiles.py)")
            return roof
        for name, v in summary["kernels"].items():
            if "k_conv3d_mid16" in name and "FETCH_SIZE_KB_avg" in v and "WRITE_SIZE_KB_avg" in v:
                roof["traffic"] = round((2.0 * v["FETCH_SIZE_KB_avg"] + v["WRITE_SIZE_KB_avg"]) * 1024.0)
                roof["traffic_source"] = os.path.relpath(files[-1], ROOT)
    except Exception as e:                                   # a malformed summary must not break the bench line
        roof["traffic_note"] = f"null: {type(e).__name__} reading the PMC summary"
    return roof


def _measure_traffic(args):
    """roofline.traffic measured IN THIS RUN: two `rocprofv3 --pmc` passes -- FETCH_SIZE and WRITE_SIZE, separate
    passes, no trace domain, exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes -- of a 3-step run of this same workload,
    as child processes of this one, started BEFORE this process touches the GPU (the program after `--` is python3 itself; the
    child returns right after its timed steps).  Returns ({"traffic": bytes per k_conv3d_mid16 launch, 2 x FETCH + WRITE, ...}, None)
    or (None, reason); a missing profiler, a refusal, a timeout or an unreadable CSV only cost the in-run number: bench.py then
    falls back to the committed summary of the same workload (_with_traffic)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="lws_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--steps", "3", "--warmup", "2", "--spinup", "0",
             "--batch", str(args.batch), "--size", args.size, "--maxdisp0", str(args.maxdisp0), "--no-cpu-baseline", "--no-pipelined",
             "--no-measure-traffic"] + (["--feature-fp16"] if args.feature_fp16 else []) + [x for o in args.opt for x in ("--opt", o)]
    kb = {}
    t0 = time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [rocprof, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "run", "--"] + child
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=args.traffic_timeout)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)               # (the exact group started above)
                except ProcessLookupError:
                    pass
                return None, f"the {ctr} pass did not finish within {args.traffic_timeout:.0f} s"
            if rc != 0:
                return None, f"rocprofv3 --pmc {ctr} exited with status {rc}"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"the {ctr} pass left no counter_collection.csv"
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if "k_conv3d_mid16" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            if not vals:
                return None, f"no k_conv3d_mid16 rows in the {ctr} pass"
            kb[ctr] = (sum(vals) / len(vals), len(vals))
    except Exception as e:                                              # noqa: BLE001 (the bench line must not depend on the profiler)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"traffic": round((2.0 * kb["FETCH_SIZE"][0] + kb["WRITE_SIZE"][0]) * 1024.0),
            "FETCH_SIZE_KB_avg": round(kb["FETCH_SIZE"][0], 2), "WRITE_SIZE_KB_avg": round(kb["WRITE_SIZE"][0], 2),
            "launches_counted": kb["FETCH_SIZE"][1], "seconds": round(time.perf_counter() - t0, 1)}, None


def _inject(point, rank):
    """Fault injection for tests/test_bench_cpu.py (first attempt only): LWS_BENCH_INJECT = "hang:R", "init-fail:R" or
    "hang-collective:R" makes rank R hang before the rendezvous, fail at it, or hang right before its first collective;
    LWS_BENCH_INJECT_ATTEMPTS=all extends it to the fallback attempt (the job then ends with the error line)."""
    spec = os.environ.get("LWS_BENCH_INJECT", "")
    if not spec or (os.environ.get("LWS_BENCH_ATTEMPT", "0") != "0" and os.environ.get("LWS_BENCH_INJECT_ATTEMPTS") != "all"):
        return
    what, _, who = spec.partition(":")
    if what != point or int(who or 0) != rank:
        return
    if what == "init-fail":
        raise RuntimeError(f"injected init failure on rank {rank} (LWS_BENCH_INJECT)")
    sys.stderr.write(f"[bench worker {rank}] injected hang at '{point}' (LWS_BENCH_INJECT)\n")
    while True:
        time.sleep(3600)


def _dry_forward(left, right):
    """CPU stand-in for LWSNet.forward in --dry-run-cpu: per-pair, no cross-sample op, four [B,1,H,W] maps."""
    d = (left - right).abs().sum(1, keepdim=True)
    return [d * (s + 1) for s in range(4)]


def dry_run(args, rank, world):
    """--dry-run-cpu: the N-rank launch, the per-rank shard of seeded pairs, the ONE gather to rank 0, the barrier-
    bracketed clock and the report -- with a CPU stand-in instead of the HIP forward.  Nothing is measured.  Like the measured
    job it runs the main leg (--batch pairs per rank per step) and, for N > 1 or --config4, config 4's 8 pairs per rank."""
    import torch.distributed as dist
    from lwsnet_amd import dist as ldist
    from lwsnet_amd.synth import make_batch
    grouped = dist.is_initialized()

    def leg(B, steps, warmup):
        left_np, right_np = make_batch(B, 16, 32, first_index=rank * B)
        left, right = torch.from_numpy(left_np), torch.from_numpy(right_np)
        # the staged gather of the measured path (lwsnet_amd.dist.StagedGather), 2 steps per gather so that full buffers, the
        # alternation of the two staging buffers and the tail flush all occur; step k "sees" pairs shifted by k so that a
        # stale or misplaced slot cannot pass the check below
        sg = ldist.StagedGather(B, 16, 32, 2, torch.device("cpu"))
        sg.warm()                                            # (as the measured path does: first-collective set-up outside the clock)
        nsteps = warmup + steps
        if grouped:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(nsteps):
            pred = _dry_forward(left + k, right)
            sg.slot().copy_(pred[3])
            sg.commit()
        sg.flush()
        if grouped:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        ok, npairs = None, 0
        if rank == 0:
            # the last gather must hold, for every rank, exactly that rank's shard of the unsharded result of the last step(s)
            al, ar = make_batch(B * world, 16, 32, first_index=0)
            al, ar = torch.from_numpy(al), torch.from_numpy(ar)
            ok = True
            for r in range(world):
                got, nvalid = sg.gathered(r)
                for j in range(nvalid):
                    k = nsteps - nvalid + j
                    want = _dry_forward(al + k, ar)[3][r * B:(r + 1) * B]
                    ok = ok and bool(torch.equal(got[j * B:(j + 1) * B], want))
                npairs += B
        return {"gather_equals_unsharded": ok, "pairs_gathered": npairs, "gathers": sg.count, "wall_s": round(elapsed, 4)}

    _inject("hang-collective", rank)
    main_leg = leg(args.batch, args.steps, args.warmup)
    c4 = leg(8, min(args.steps, 4), 1) if (world > 1 or args.config4) else None
    ok = True
    if rank == 0:
        out = {"metric": "stereo pairs/sec @256x512 maxdisp=192 (stage-4)", "value": None, "unit": "pairs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, **main_leg,
               "collective": {"backend": dist.get_backend() if grouped else None, "world": world},
               "note": "CPU/gloo plumbing check with a stand-in forward; no GPU, nothing measured"}
        if c4 is not None:
            out["config4"] = {"workload": f"BASELINE config 4 (dry run): 8 pairs/GPU x {world} ranks = batch {8 * world}",
                              "pairs_per_gpu": 8, "global_batch": 8 * world, "pairs_per_s": None, "ms_per_step": None,
                              "all_ranks_slots_equal_unsharded": c4["gather_equals_unsharded"], **c4}
        print(json.dumps(out), flush=True)
        ok = bool(main_leg["gather_equals_unsharded"]) and (c4 is None or bool(c4["gather_equals_unsharded"]))
    if grouped:
        dist.destroy_process_group()
    if rank == 0 and not ok:
        raise SystemExit(1)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="pairs per GPU per step (BASELINE config 2: 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the untimed lws_pool throughput extra")
    ap.add_argument("--pool-workers", type=int, default=4, help="worker threads of the lws_pool extra")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent handles/HIP streams the steps rotate over (1 = every step on one stream; >1 "
                         "overlaps consecutive steps: higher pairs/s, but kernels then share CUs and their in-situ "
                         "durations -- hence roofline.frac -- grow)")
    ap.add_argument("--size", default="256x512", help="HxW of the synthetic pairs (default: BASELINE config 2)")
    ap.add_argument("--feature-fp16", action="store_true", help="BASELINE config 5: fp16-rounded feature maps")
    ap.add_argument("--maxdisp0", type=int, default=24, help="stage-1 hypotheses (24 = maxdisp 192, 32 = maxdisp 256)")
    ap.add_argument("--spinup", type=float, default=0.3, help="seconds of untimed forwards before the warm-up steps (clock ramp of an idle GPU)")
    ap.add_argument("--gather-pairs", type=int, default=None,
                    help="minimum pairs per rank carried by one RCCL gather (staged gather); default: lwsnet_amd.dist.gather_policy(world size)")
    ap.add_argument("--opt", action="append", default=[], help="name=value launch-plan option (lws_set_option); experiments only")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="plumbing check without a GPU: gloo, per-pair stand-in forward, value = null (tests only)")
    ap.add_argument("--one-gpu", action="store_true",
                    help="N > 1 ranks of the REAL HIP path sharing cuda:0 (gloo; the gather carries device maps through the host: "
                         "RCCL refuses duplicate devices).  Exercises the N-rank code with two processes on one GPU; the ranks share "
                         "the chip, so nothing it prints is a throughput: value = null (tests/test_gpu_dist.py)")
    ap.add_argument("--collective", choices=("rccl", "gloo-host"), default="rccl",
                    help="how the stage-4 maps reach rank 0: one RCCL gather over xGMI (default), or gloo through host memory -- "
                         "the labelled fallback the supervisors start when the RCCL job fails (lwsnet_amd/launch.py)")
    ap.add_argument("--job-timeout", type=float, default=600.0,
                    help="seconds one attempt of the job may take before its workers are killed and the fallback (or the error "
                         "line) takes over; also the in-process deadline of a plain single-GPU run")
    ap.add_argument("--init-timeout", type=float, default=120.0,
                    help="seconds for init_process_group (and, through torch's NCCL watchdog, every collective)")
    ap.add_argument("--config4", action="store_true",
                    help="also time BASELINE config 4's per-rank shape (8 pairs per GPU per step) and report it under `config4`; "
                         "on by itself for N > 1")
    ap.add_argument("--no-measure-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic in this run (single-GPU runs only); "
                         "the committed PMC summary of the same workload is reported instead")
    ap.add_argument("--traffic-timeout", type=float, default=60.0, help="seconds one rocprofv3 --pmc child pass may take")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)      # the child of _measure_traffic: timed steps only
    ap.add_argument("--config4-steps", type=int, default=None, help="timed steps of the config-4 leg (default: 10..50 following --steps)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    from lwsnet_amd import launch
    if os.environ.get("LWS_BENCH_WORKER") != "1":
        if "WORLD_SIZE" in os.environ and "RANK" in os.environ:
            # a torchrun rank (the driver's N > 1 launch, or the child of the branch below): supervise a fresh worker
            if int(os.environ["WORLD_SIZE"]) != args.gpus:
                raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch with torch.distributed.run for N > 1")
            raise SystemExit(launch.supervise(os.path.abspath(__file__), sys.argv[1:], args))
        if args.gpus > 1:
            raise SystemExit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus, args.job_timeout, args.steps, args.warmup))
        # plain single-GPU run, in this process: an in-process deadline instead of a supervisor
        launch.arm_watchdog(args.job_timeout, lambda: print(launch.error_line(
            1, args.steps, args.warmup, f"the run did not finish within --job-timeout {args.job_timeout:.0f} s"), flush=True))
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)      # (this process itself runs under rocprofv3)
        if not (args.no_measure_traffic or args.traffic_child or args.dry_run_cpu or profiled):
            args.traffic_in_run = _measure_traffic(args)            # children first: this process has not touched the GPU yet
    try:
        return worker(args)
    except (Exception, SystemExit) as e:
        # a worker that fails says why on stdout too, so that its supervisor can put the reason into the job's one line
        if os.environ.get("LWS_BENCH_WORKER") == "1" and not (isinstance(e, SystemExit) and e.code in (0, None)):
            print(launch.error_line(args.gpus, args.steps, args.warmup,
                                    f"rank {os.environ.get('RANK', '0')}: {type(e).__name__}: {str(e)[:300]}"), flush=True)
        raise


def worker(args):
    from lwsnet_amd import _lib, dist as ldist
    from lwsnet_amd.models import LWSNet
    from lwsnet_amd.synth import make_batch
    from lwsnet_amd.weights import default_args, make_state_dict

    global H, W
    H, W = [int(v) for v in args.size.split("x")]
    env_rank = int(os.environ.get("RANK", "0"))
    _inject("hang", env_rank)
    _inject("init-fail", env_rank)
    through_host = args.dry_run_cpu or args.one_gpu or args.collective == "gloo-host"
    rank, local_rank, world = ldist.init_from_env("gloo" if through_host else None, timeout_s=args.init_timeout)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if args.dry_run_cpu:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the measured path)")
    if not args.one_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"{torch.cuda.device_count()} HIP device(s) visible, the job has {world} ranks (one device per rank)")
    dev = ldist.local_device(local_rank, share_one_gpu=args.one_gpu)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    # scalars that cross ranks (MAX of the elapsed time): on the device for RCCL, on the host for gloo
    red_dev = dev if (dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")

    margs = default_args(maxdisplist=(args.maxdisp0, 5, 5), feature_fp16=args.feature_fp16)
    sd = make_state_dict(7, margs)
    c3_first = margs.channels_3d * margs.growth_rate[0]
    S = max(1, args.streams)
    models = [LWSNet(margs, device=dev).set_state_dict(sd).eval() for _ in range(S)]
    for o in args.opt:
        for m in models:
            m.set_option(o.split("=")[0], int(o.split("=")[1]))
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [None]
    model = models[0]
    lib = _lib.load()
    grouped = dist.is_initialized()                      # torchrun launch (any world size, also 1): run the collective
    KC_MID16, KC_CONV64 = 3, 11          # LWS_KC_CONV3D_MID16, LWS_KC_REF_CONV64 (include/lwsnet_hip.h)
    counter = [0]

    def max_over_ranks(x):
        if not grouped:
            return float(x)
        t = torch.tensor([x], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run_leg(B, steps, warmup, spinup_s=0.0, before_timed=None, after_timed=None):
        """W warm-up steps, then exactly K timed steps of B pairs per rank bracketed by barrier + synchronize on both sides; with a
        process group every step's stage-4 map goes into the staged gather and the tail is flushed before the clock stops.  Then
        the checks (rank 0's slot == its own last map; every other rank's slot == what rank 0 computes for that rank's pairs) and
        the same K steps without the gather (what the ONE collective costs).  Times are the MAX over ranks."""
        left_np, right_np = make_batch(B, H, W, first_index=rank * B)
        left, right = torch.from_numpy(left_np).to(dev), torch.from_numpy(right_np).to(dev)
        for m in models:
            _lib.check(lib.lws_reserve(m._h, B, H, W), "lws_reserve")
        # The ONE collective of the path: stage-4 maps -> rank 0 (north_star: "a single RCCL gather for
        # the output disparities" of a batch).  The stage-4 maps of consecutive steps are written straight into the slots of a
        # staging buffer (lws_forward's output pointer: no copy) and gathered together; how many pairs per rank one gather carries
        # is a function of the world size (lwsnet_amd.dist.gather_policy: 8 pairs, 16 on eight ranks), chosen from the root-rank
        # emulation of round 4 (profiles/r04/gather_root_emulation_d.txt).  Two staging buffers alternate, so
        # the asynchronous gather of one overlaps the forwards that fill the other; every step's map is gathered inside the timed
        # region (the tail is flushed before the clock stops).  NOTE (r04): in a world of ONE torch's NCCL gather is a tensor copy
        # of the root's own shard -- no RCCL kernel runs -- so `collective.overhead_pct` of a one-rank run prices the staging, the
        # stream hand-offs and that copy only.
        G = ldist.gather_every(world, B, args.gather_pairs) if grouped else 1
        sg = ldist.StagedGather(B, H, W, G, dev, multi_stream=S > 1) if grouped else None

        def step(gather=True):
            i = counter[0] % S
            counter[0] += 1
            out = [None, None, None, sg.slot()] if (grouped and gather) else None
            if S == 1:
                pred = models[0](left, right, out=out)
                if out is not None:
                    sg.commit()
                return pred
            with torch.cuda.stream(streams[i]):
                pred = models[i](left, right, out=out)
                if out is not None:
                    sg.commit()
            return pred

        # spin-up (untimed, before the W warm-up steps): a short run started on an idle GPU measures the clock ramp, not the
        # path -- the driver's --steps 20 --warmup 5 is 13 ms of GPU work and read 3.5 % below --steps 200 (r03).  Forwards for
        # spinup_s seconds first -- at least that long, then until two consecutive blocks of 25 forwards take the same time within
        # 1 % (<= 2 s in all); then W warm-up steps; then exactly K timed steps.
        t_spin = time.perf_counter()
        n_spin, prev, settled = 0, None, False
        while spinup_s > 0 and (time.perf_counter() - t_spin < spinup_s or (not settled and time.perf_counter() - t_spin < 2.0)):
            t1 = time.perf_counter()
            for _ in range(25):
                models[n_spin % S](left, right)
                n_spin += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            settled = prev is not None and abs(dt - prev) <= 0.01 * prev
            prev = dt
        spun_s = time.perf_counter() - t_spin
        if grouped:
            _inject("hang-collective", rank)
            sg.warm()                        # the communicator's one-time set-up must not land in the timed region
        for _ in range(warmup):
            step()
        if grouped:
            sg.flush()                       # ... nor a half-filled buffer of warm-up steps
            sg.reset()
        if before_timed:
            before_timed()
        if grouped:
            sg.count = 0
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pred = step()
        if grouped:
            sg.flush()                                       # tail gather + wait for everything in flight
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if after_timed:
            after_timed()
        gather_ok = all_ok = None
        if grouped and rank == 0:                            # rank 0's part of the last gather holds this rank's last stage-4 map
            mine, nvalid = sg.gathered(0)
            gather_ok = bool(torch.equal(mine[(nvalid - 1) * B:nvalid * B], pred[3]))
            # ... and every other rank's part must be what THIS rank computes for that rank's pairs (pure batch sharding: the
            # sharded job equals the unsharded forward bit for bit) -- untimed, world forwards on rank 0
            all_ok = gather_ok
            for r in range(1, world):
                ql, qr = make_batch(B, H, W, first_index=r * B)
                want = models[0](torch.from_numpy(ql).to(dev), torch.from_numpy(qr).to(dev))[3]
                theirs, nv = sg.gathered(r)
                all_ok = all_ok and nv == nvalid and bool(torch.equal(theirs[(nv - 1) * B:nv * B], want))
        # what the ONE collective of the path costs ("<= 3 %"): the same K steps again without the
        # gather, same barriers, same clock -- measurable on one GPU under `torch.distributed.run --nproc-per-node 1`
        t_plain = None
        if grouped:
            def plain_steps(n):
                dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n):
                    step(gather=False)
                torch.cuda.synchronize()
                dist.barrier()
                return time.perf_counter() - t1

            plain_steps(min(warmup, 3))
            t_plain = max_over_ranks(plain_steps(steps))
        return {"B": B, "steps": steps, "elapsed": max_over_ranks(elapsed), "pred": pred, "left": left, "right": right,
                "left_np": left_np, "right_np": right_np, "G": G, "gathers": sg.count if grouped else 0, "spun_s": spun_s,
                "gather_ok": gather_ok, "all_ok": all_ok, "t_plain": t_plain}

    # inside the timed region only the dominant kernel class is bracketed by hipEvents (8 events per sampled step), and only
    # on every n-th step; the per-class breakdown comes from a separate, untimed pass below.  A kernel bracketed by its own
    # events keeps its successor from being queued behind it: ~3 us per timed launch at batch 1 unprofiled, 8 us under a
    # kernel trace (profiles/r04/timeline_b1_trace_only.txt shows such a forward), so timing EVERY step of a short run taxed
    # the headline by 3-4 %.  Every 8th step (round 4; every 4th before): the driver's --steps 20 gives 3 sampled
    # steps = 12 timed launches, the default 50 steps 7 = 28; the durations spread by +-1 %.
    sample_every = max(8, (args.steps // S) // 12)
    tot = (ctypes.c_double * _lib.LWS_KC_COUNT)()
    cnt = (ctypes.c_int64 * _lib.LWS_KC_COUNT)()
    mid = {"ms": 0.0, "n": 0}

    def arm_profiler():
        for m in models:
            _lib.check(lib.lws_profile_enable(m._h, 1 << KC_MID16), "lws_profile_enable")
            _lib.check(lib.lws_profile_sample(m._h, sample_every), "lws_profile_sample")

    def read_profiler():
        for m in models:
            _lib.check(lib.lws_profile_read(m._h, tot, cnt), "lws_profile_read")
            _lib.check(lib.lws_profile_enable(m._h, 0), "lws_profile_enable")
            mid["ms"] += tot[KC_MID16]
            mid["n"] += cnt[KC_MID16]

    B = args.batch
    leg = run_leg(B, args.steps, args.warmup, spinup_s=max(0.0, args.spinup), before_timed=arm_profiler, after_timed=read_profiler)
    if args.traffic_child:
        return                               # (under rocprofv3 --pmc: the counters of the launches above are all that is wanted)
    elapsed, pred, left, right, left_np, right_np = leg["elapsed"], leg["pred"], leg["left"], leg["right"], leg["left_np"], leg["right_np"]
    G, spun_s, gather_ok, all_ok = leg["G"], leg["spun_s"], leg["gather_ok"], leg["all_ok"]
    # the clock the dominant kernel held on this box / on every rank's GPU (lws_clock_stamp: s_memtime against s_memrealtime
    # inside k_conv3d_mid16): 16 more forwards right after the timed region -- same queue depth, same mix of kernels -- stamped;
    # the last stage-1 launch of the last forward is read
    clock_ghz = None
    if c3_first != 8 and not (model.get_option("split_bf16") & 1):      # (k_conv3d_mid16x, the split-bf16 form, carries no stamps)
        ghz = ctypes.c_double(0.0)
        _lib.check(lib.lws_clock_stamp(model._h, 1), "lws_clock_stamp")
        for _ in range(16):
            models[0](left, right)
        _lib.check(lib.lws_clock_read(model._h, ctypes.byref(ghz)), "lws_clock_read")
        _lib.check(lib.lws_clock_stamp(model._h, 0), "lws_clock_stamp")
        clock_ghz = float(ghz.value)
    clock_per_rank = None
    if grouped and world > 1:
        t = torch.tensor([clock_ghz or 0.0], device=red_dev, dtype=torch.float64)
        allc = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allc, t)
        clock_per_rank = [round(float(c.item()), 4) for c in allc]
    mid_ms, mid_n = mid["ms"], mid["n"]
    mid_avg_us = 1e3 * mid_ms / max(mid_n, 1)
    collective_overhead = None
    if grouped and not through_host:
        collective_overhead = {"ms_per_step_without_gather": round(1e3 * leg["t_plain"] / args.steps, 4)}
    # BASELINE config 4's per-rank shape in the same job: 8 pairs per GPU per step (batch 8 N in all), its own warm-up, clock,
    # staged gather, bit-equality check and gather price.  `value` above stays the 1-pair-per-GPU number.
    config4 = None
    if (world > 1 or args.config4) and (H, W, args.maxdisp0) == (256, 512, 24) and not args.feature_fp16:
        k4 = args.config4_steps if args.config4_steps else max(10, min(args.steps, 50))
        l4 = leg if B == 8 else run_leg(8, k4, max(3, min(args.warmup, 10)))
        ms4 = 1e3 * l4["elapsed"] / l4["steps"]
        config4 = {"workload": f"BASELINE config 4: {world} x MI355X batch-sharded, batch={8 * world} (8 pairs/GPU), 256x512 synthetic, "
                               "maxdisplist=[24,5,5]",
                   "pairs_per_gpu": 8, "global_batch": 8 * world, "steps": l4["steps"],
                   "pairs_per_s": None if args.one_gpu else round(world * 8 * l4["steps"] / l4["elapsed"], 2), "ms_per_step": round(ms4, 4),
                   "gather_every_steps": l4["G"], "gathers_in_timed_region": l4["gathers"],
                   "rank0_slot_equals_local": l4["gather_ok"], "all_ranks_slots_equal_unsharded": l4["all_ok"]}
        if grouped and l4["t_plain"] is not None and not through_host:
            base4 = 1e3 * l4["t_plain"] / l4["steps"]
            config4["ms_per_step_without_gather"] = round(base4, 4)
            config4["gather_overhead_pct"] = round(100.0 * (ms4 - base4) / base4, 2)
        if args.one_gpu:
            config4["pairs_per_s_all_ranks_on_one_gpu"] = round(world * 8 * l4["steps"] / l4["elapsed"], 2)
        if B != 8:                           # back to the main leg's shape for the untimed passes below
            for m in models:
                _lib.check(lib.lws_reserve(m._h, B, H, W), "lws_reserve")
    # untimed breakdown pass: every kernel class, 10 steps
    nb = 10
    KC_MID8 = 4
    _lib.check(lib.lws_profile_enable(model._h, -1), "lws_profile_enable")
    for _ in range(nb):
        models[0](left, right)
    torch.cuda.synchronize()
    _lib.check(lib.lws_profile_read(model._h, tot, cnt), "lws_profile_read")
    mid8_ms = (ctypes.c_float * 4096)()
    mid8_n = ctypes.c_int(0)
    _lib.check(lib.lws_profile_read_class(model._h, KC_MID8, mid8_ms, 4096, ctypes.byref(mid8_n)), "lws_profile_read_class")
    mid8_each = [mid8_ms[i] for i in range(min(mid8_n.value, 4096))]
    _lib.check(lib.lws_profile_enable(model._h, 0), "lws_profile_enable")
    # untimed latency pass: one forward at a time, host call -> result complete (SURVEY.md section 8d asks for the
    # spread as well as the mean; the throughput above keeps the stream full, this does not)
    lat = []
    for _ in range(50):
        t1 = time.perf_counter()
        models[0](left, right)
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t1))
    lat.sort()
    latency = {"p10": round(lat[5], 4), "p50": round(lat[25], 4), "p90": round(lat[45], 4),
               "what": "ms per isolated forward (host call to stream idle), 50 samples"}
    # untimed extra (single GPU, default single-stream run only): batch-1 forwards issued by 3 host threads on 3 handles / HIP
    # streams, so that the launch-latency-bound chains of consecutive forwards overlap.  Reported beside `value`, never as it:
    # kernels of different forwards then share the CUs, so per-kernel durations (and roofline.frac) are not comparable.
    pipelined = None
    if not grouped and S == 1 and not args.no_pipelined:
        P = args.pool_workers
        per = max(args.steps, 200) * P
        runs = []
        same = True
        pool_mid = None
        with model.pool(workers=P) as pool:
            pool.reserve(B, H, W)
            outs = [[torch.empty((B, 1, H, W), device=dev) for _ in range(4)] for _ in range(2 * P)]
            for rep in range(5):                       # first repetition = warm-up; three timed ones show the spread; the
                if rep == 4:                           # fifth (untimed) samples k_conv3d_mid16 INSIDE the pool for its roofline
                    pool.wait_all()
                    pool.profile(1 << KC_MID16, 4)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                jobs = []
                for k in range(per):
                    if len(jobs) >= 2 * P:
                        jobs.pop(0).result()
                    jobs.append(pool.submit(left, right, out=outs[k % (2 * P)]))
                last = [j.result() for j in jobs][-1]
                dtp = time.perf_counter() - t1
                if 1 <= rep <= 3:
                    runs.append(B * per / dtp)
                same = same and all(bool(torch.equal(a, b)) for a, b in zip(last, pred))
                if rep == 4:
                    pool.wait_all()
                    ptot, pcnt = pool.profile_read()
                    pool.profile(0)
                    if pcnt[KC_MID16]:
                        pool_mid = (1e3 * ptot[KC_MID16] / pcnt[KC_MID16], int(pcnt[KC_MID16]), B * per / dtp)
        runs.sort()
        pipelined = {"value": round(runs[len(runs) // 2], 2), "unit": "pairs/s", "workers": P, "steps": per,
                     "min": round(runs[0], 2), "max": round(runs[-1], 2), "spread_pct": round(100.0 * (runs[-1] - runs[0]) / runs[len(runs) // 2], 2),
                     "ms_per_step": round(1e3 * B / runs[len(runs) // 2], 4), "outputs_equal_single_stream": same,
                     "_pool_mid": pool_mid,
                     "what": f"lws_pool (C ABI): {P} C++ worker threads, each with a clone of the model and ONE HIP stream, keep "
                             f"{2 * P} batch-{B} forwards in flight so that their launch-bound chains overlap on the device; median of "
                             "3 timed repetitions; not the headline: `value` is the single-stream number"}
    # untimed extra (single GPU, default single-stream run only): the same steps with option split_bf16 = 1 -- the Conv3D middle
    # layers and refinement2[0] on split-bf16 MFMA (k_conv3d_mid16x, k_conv3d_mid8x, k_ref_conv64x: three bf16 values per
    # float32 operand, six exact cross products accumulated in float32).  An opt-in numerics mode: float32-level accuracy (tests/test_gpu_parity.py::
    # test_split_bf16_*), NOT bit-exact against the oracle chain, therefore never `value` and reported with its own dtype.
    split_bf16 = None
    if not grouped and S == 1 and not args.no_pipelined and model.get_option("split_bf16") == 0 and c3_first == 32:
        model.set_option("split_bf16", 7)
        try:
            for _ in range(10):
                px = model(left, right)
            _lib.check(lib.lws_profile_enable(model._h, (1 << KC_MID16) | (1 << KC_CONV64)), "lws_profile_enable")
            _lib.check(lib.lws_profile_sample(model._h, 8), "lws_profile_sample")
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nx = max(args.steps, 100)
            for _ in range(nx):
                px = model(left, right)
            torch.cuda.synchronize()
            dtx = time.perf_counter() - t1
            totx = (ctypes.c_double * _lib.LWS_KC_COUNT)()
            cntx = (ctypes.c_int64 * _lib.LWS_KC_COUNT)()
            _lib.check(lib.lws_profile_read(model._h, totx, cntx), "lws_profile_read")
            _lib.check(lib.lws_profile_enable(model._h, 0), "lws_profile_enable")
            x_us = 1e3 * totx[KC_MID16] / max(cntx[KC_MID16], 1)
            x64_us = 1e3 * totx[KC_CONV64] / max(cntx[KC_CONV64], 1)
            diff = [round(float((px[s_] - pred[s_]).abs().max()), 6) for s_ in range(4)]
            split_bf16 = {"value": round(B * nx / dtx, 2), "unit": "pairs/s", "steps": nx, "ms_per_step": round(1e3 * dtx / nx, 4),
                          "dtype": "f32 activations / weights split into 3 x bf16 for the MFMAs of the Conv3D middle layers and of "
                                   "refinement2[0], f32 accumulate",
                          "k_conv3d_mid16x_avg_launch_us": round(x_us, 2), "k_ref_conv64x_avg_launch_us": round(x64_us, 2),
                          "max_abs_vs_exact_per_stage": diff,
                          "what": "option split_bf16 = 7 (k_conv3d_mid16x, k_conv3d_mid8x, k_ref_conv64x): not bit-exact against the oracle chain (float32-level accuracy, gated by "
                                  "the float64 noise-floor tests); an opt-in numerics mode, never the headline"}
        finally:
            model.set_option("split_bf16", 0)
    kernels = {}
    for kc in range(_lib.LWS_KC_COUNT):
        if cnt[kc]:
            kernels[lib.lws_kernel_class_name(kc).decode()] = {"launches_per_step": cnt[kc] / nb,
                                                               "avg_us": 1e3 * tot[kc] / cnt[kc]}
    hot_ms = sum(tot[kc] for kc in range(8)) / nb
    all_ms = sum(tot) / nb

    # dominant kernel: stage-1 Conv3D c3 -> c3 (k_conv3d_mid16): 2*27*c3*c3 FLOP per voxel, voxels = B*D1*(H/8)*(W/8)
    c3 = margs.channels_3d * margs.growth_rate[0]
    h2_, w2_ = (H + 1) // 2, (W + 1) // 2                    # the stem gives ceil(H/2); the hourglass halves twice more
    # pairs per launch: read off the launch count of the breakdown pass rather than assumed
    n_mid16 = kernels.get("conv3d_mid16", {}).get("launches_per_step", margs.layers_3d)
    pairs_per_launch = B * margs.layers_3d / max(n_mid16, 1)
    vox = pairs_per_launch * margs.maxdisplist[0] * (h2_ // 4) * (w2_ // 4)
    flop_per_launch = 2.0 * 27 * c3 * c3 * vox
    # algorithmic FLOPs of one forward: Conv3D stacks + 2D feature extractor (both images) + refinement
    L3 = margs.layers_3d
    vox_s = [margs.maxdisplist[0] * (h2_ // 4) * (w2_ // 4), (2 * margs.maxdisplist[1] - 1) * (h2_ // 2) * (w2_ // 2),
             (2 * margs.maxdisplist[2] - 1) * h2_ * w2_]
    c3_s = [margs.channels_3d * g for g in margs.growth_rate]
    gf_conv3d = sum(2.0 * 27 * v * (2 * c + L3 * c * c) for v, c in zip(vox_s, c3_s)) / 1e9
    gf_feat = 0.486 * (H * W) / (256.0 * 512.0)              # 22 small layers, both images
    gf_ref = (2.0 * 9 * 32 * 4 + 12 * 2.0 * (9 * 32 + 32 * 32) + 2.0 * 9 * 64 * 32 + 2.0 * 9 * 32) * H * W / 1e9
    gf_pair = gf_conv3d + gf_feat + gf_ref
    mid = {"avg_us": mid_avg_us} if mid_avg_us > 0 else None
    roof = None
    if mid:
        achieved = flop_per_launch / (mid["avg_us"] * 1e-6) / 1e12
        # with --opt split_bf16=1 (experiments only) the class runs k_conv3d_mid16x: six bf16 MFMAs per float32 product,
        # so the ceiling for the same algorithmic FLOPs is the dense bf16 peak / 6
        split = (model.get_option("split_bf16") & 1) != 0
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if split else PEAK_F32_MFMA_TFLOPS
        roof = {"bound": "mfma", "kernel": "k_conv3d_mid16x<3,4> (split-bf16, NOT the oracle chain)" if split else "k_conv3d_mid16<32,3,4>",
                "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "peak_note": "bf16 dense peak / 6 cross products" if split else
                             "nominal: 64 FLOP/clk/SIMD at 2.4 GHz; see clock_ghz for the clock this kernel held in this run",
                "clock_ghz": round(clock_ghz, 4) if clock_ghz else None,
                # the same achieved rate against the peak at the clock the kernel held (peak x clock / 2.4 GHz)
                "frac_at_held_clock": round(achieved / (peak * clock_ghz / 2.4), 4) if clock_ghz else None,
                **({"clock_ghz_per_rank": clock_per_rank} if clock_per_rank else {}),
                "clock_note": "in-kernel clock of k_conv3d_mid16 (lws_clock_stamp: d s_memtime / d s_memrealtime x 100 MHz, median of "
                              "64 workgroups of the last stage-1 launch of 16 forwards issued right after the timed region); peak is "
                              "priced at 2.4 GHz",
                "traffic": None, "flop_per_launch": flop_per_launch, "avg_launch_us": round(mid["avg_us"], 2),
                "timed_launches": int(mid_n), "timed_every_nth_step": sample_every, "pairs_per_launch": pairs_per_launch}
    if pipelined is not None:
        pm = pipelined.pop("_pool_mid", None)
        if pm is not None:
            # k_conv3d_mid16 as it runs INSIDE the pool, beside the other workers' kernels (event-bracketed launches of a
            # fifth, untimed repetition, every 4th forward of every worker): the same algorithmic FLOPs per launch
            p_ach = flop_per_launch / (pm[0] * 1e-6) / 1e12
            pipelined["roofline"] = {"bound": "mfma", "kernel": "k_conv3d_mid16<32,3,4> beside the other workers' kernels",
                                     "achieved": round(p_ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(p_ach / PEAK_F32_MFMA_TFLOPS, 4), "avg_launch_us": round(pm[0], 2),
                                     "timed_launches": pm[1], "pairs_per_s_while_sampling": round(pm[2], 2),
                                     "step_frac": round(gf_pair * pipelined["value"] / 1e3 / PEAK_F32_MFMA_TFLOPS, 4),
                                     "note": "in-pool launch duration from hipEvent pairs around the launch (not kernel timestamps): "
                                             "co-resident kernels of other forwards stretch it; step_frac = algorithmic GF of a "
                                             "forward x pairs/s against the same peak"}
    # the whole step against the same fp32-MFMA peak: algorithmic GF of a forward x pairs / step time.  This, not `frac`,
    # is how far the PATH is from the roofline (batch 1: 35 dependent launches, ~40 % of the step is fixed launch cost).
    step_tf = gf_pair * B / (1e3 * elapsed / args.steps) if elapsed > 0 else 0.0      # GF per ms = TF
    if roof:
        roof["step_frac"] = round(step_tf / PEAK_F32_MFMA_TFLOPS, 4)
        roof["step_achieved"] = round(step_tf, 2)
        roof["step_gflop_per_pair"] = round(gf_pair, 3)
    # second MFMA kernel of the path: the 8 -> 8 Conv3D layers of stages 2 and 3 (8 launches per step), from the untimed
    # breakdown pass (events around each launch, so each carries ~1-2 us of dispatch): USEFUL FLOPs / launch time
    secondary = None
    if mid8_each and len(mid8_each) % (2 * L3 * nb) == 0 and c3_s[1] == 8 and c3_s[2] == 8:
        per_stage = {}
        ppl8 = B * (2 * L3 * nb) / len(mid8_each)              # pairs per launch (B, or B/2 under the batch split)
        for si, name in ((1, "stage2"), (2, "stage3")):
            us = [1e3 * mid8_each[k] for k in range(len(mid8_each)) if (k % (2 * L3)) // L3 == si - 1]
            gf = 2.0 * 27 * 8 * 8 * vox_s[si] * ppl8 / 1e9
            avg = sum(us) / len(us)
            per_stage[name] = {"avg_launch_us": round(avg, 2), "useful_gflop_per_launch": round(gf, 4),
                               "achieved": round(gf / avg * 1e3, 2), "frac": round(gf / avg * 1e3 / PEAK_F32_MFMA_TFLOPS, 4)}
        tot_us = sum(1e3 * v for v in mid8_each) / nb
        tot_gf = sum(2.0 * 27 * 64 * vox_s[si] * B * L3 for si in (1, 2)) / 1e9
        secondary = {"kernel": "k_conv3d_mid8q (3x2x32 / 3x8x32 tiles by grid size)",
                     "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F32_MFMA_TFLOPS,
                     "launches_per_step": len(mid8_each) // nb, "pairs_per_launch": ppl8, "us_per_step": round(tot_us, 2),
                     "achieved": round(tot_gf / tot_us * 1e3, 2), "frac": round(tot_gf / tot_us * 1e3 / PEAK_F32_MFMA_TFLOPS, 4), **per_stage,
                     "note": "useful FLOPs only; event pairs around each launch (untimed pass), not kernel timestamps"}

    # HBM-bound kernels of the path: ALGORITHMIC bytes per launch / average launch duration of the
    # untimed breakdown pass above, against the 8 TB/s peak.  (Batch 1: latency-bound launches of a few MB.)
    hbm = {}
    h2, w2 = (H + 1) // 2, (W + 1) // 2
    warp_bytes = [(2 * 16 * (h2 // 2) * (w2 // 2) + 9 * (h2 // 2) * (w2 // 2) + (h2 // 2) * (w2 // 2)) * 4 * B,
                  (2 * 8 * h2 * w2 + 9 * h2 * w2 + h2 * w2) * 4 * B]
    # (per-step byte totals over the launches of a step: the refinement runs in chunks of pairs -- option ref_chunk_mb --, so a
    # launch of k_ref_dws covers one chunk, not the batch)
    # Conv3D first / last layers ("un-fused activation traffic: each layer read-in + write-out"): the
    # first layer reads the one-channel volume (stage 1 inside a forward: the two 16-channel feature maps, and writes the raw
    # volume too) and writes C3 channels per voxel; the last layer reads C3 channels + the skip value per voxel and writes the
    # volume (stage 1) or, with the soft-argmin fused (stages 2-3), one low-resolution map
    hw_s = [(h2 // 4) * (w2 // 4), (h2 // 2) * (w2 // 2), h2 * w2]
    first_bytes = sum(((2 * 16 * hw_s[0] + vox_s[0]) if si == 0 else vox_s[si]) + c3_s[si] * vox_s[si] for si in range(3)) * 4.0 * B
    last_bytes = sum((c3_s[si] + 1) * vox_s[si] + (vox_s[si] if si == 0 else hw_s[si]) for si in range(3)) * 4.0 * B
    # (refinement2's last block runs inside k_ref_dws_last -- class ref_last -- at batch 1: option "fuse_ref_last")
    frl = model.get_option("fuse_ref_last")
    dws_blocks = 11 if (frl == 1 or (frl == -1 and B <= 1)) else 12
    # (the first block of each refinement1 branch reads its 1- or 3-plane input instead of a 32-channel map when the branch's first
    # convolution runs inside it: option "fuse_first")
    ff = model.get_option("fuse_first")
    dws_bytes = (dws_blocks * 2.0 * 32 - ((32 - 1) if ff & 1 else 0) - ((32 - 3) if ff & 2 else 0)) * B * H * W * 4
    for name, step_bytes in (("volume_l1_warp", float(sum(warp_bytes))), ("ref_dws", dws_bytes),
                             ("softargmin", (margs.maxdisplist[0] * (h2 // 4) * (w2 // 4) + H * W) * 4.0 * B),
                             ("conv3d_first", first_bytes), ("conv3d_last", last_bytes)):
        if name in kernels:
            nl = kernels[name]["launches_per_step"]
            nbytes = step_bytes / nl
            gbs = nbytes / (kernels[name]["avg_us"] * 1e-6) / 1e9
            hbm[name] = {"algorithmic_bytes_per_launch": int(nbytes), "launches_per_step": nl, "avg_us": round(kernels[name]["avg_us"], 2),
                         "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000.0, 4)}

    if rank != 0:
        if grouped:
            dist.destroy_process_group()
        return

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import lws_oracle                      # checker only: the CPU leg of the report
        l1, r1 = left_np[:1], right_np[:1]
        # torch-CPU oversubscribes badly on many-core hosts (256 threads: 47 s per pair): probe a few thread
        # counts with one forward each and keep the fastest, then time 3 forwards with it
        ncpu = os.cpu_count() or 1
        best_t, best_n = None, 1
        for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(n)
            lws_oracle.forward(l1, r1, sd, margs.maxdisplist)     # warm-up for this thread count
            t1 = time.perf_counter()
            lws_oracle.forward(l1, r1, sd, margs.maxdisplist)
            dt = time.perf_counter() - t1
            if best_t is None or dt < best_t:
                best_t, best_n = dt, n
            if dt > 8.0:
                break
        torch.set_num_threads(best_n)
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            ref = lws_oracle.forward(l1, r1, sd, margs.maxdisplist)
            ts.append(time.perf_counter() - t1)
        med = sorted(ts)[1]
        err = [float((pred[s][:1].cpu() - ref[s]).abs().max()) for s in range(4)]
        from lwsnet_amd.metrics import error_3px
        e3 = error_3px(pred[3][:1].cpu().numpy(), np.maximum(ref[3].numpy(), 1e-3), 192)   # finetune.py:212-219, oracle as GT
        cpu = {"value": round(1.0 / med, 3), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"3 forwards of 1 pair {H}x{W}, median, after probing 8/16/32/64 threads (host has {ncpu} logical "
                         f"CPUs, cgroup cpu.max '{_cpu_quota()}'); literal oracle on torch-CPU {torch.__version__} (Paddle-CPU stand-in)",
               "max_abs_vs_gpu_per_stage": [round(e, 6) for e in err], "err_3px_stage4_vs_oracle": e3}
        # Numerics account (untimed): how far the GPU result and the float32 literal oracle each sit from the float64
        # literal oracle, per stage -- the smooth bench pair and a white-noise pair (the adversarial case).
        # north_star's 1e-3 px at stage 4 is below the float32 noise floor of the reference algorithm itself; the
        # comparable statement is gpu_vs_fp64 <= ~1x literal_fp32_vs_fp64.
        from lwsnet_amd.synth import make_noise_pair

        def dist64(p, r64):
            return [round(float((p[s].double() - r64[s]).abs().max()), 6) for s in range(4)]

        ref64 = lws_oracle.forward(l1, r1, sd, margs.maxdisplist, dtype=torch.float64)
        cpu["max_abs_gpu_vs_fp64"] = dist64([pred[s][:1].cpu() for s in range(4)], ref64)
        cpu["max_abs_literal_fp32_vs_fp64"] = dist64(ref, ref64)
        nl, nr = make_noise_pair(H, W, 0)
        npred = [p.cpu() for p in model(nl[None], nr[None])]
        n32 = lws_oracle.forward(nl[None], nr[None], sd, margs.maxdisplist)
        n64 = lws_oracle.forward(nl[None], nr[None], sd, margs.maxdisplist, dtype=torch.float64)
        cpu["noise_pair"] = {"max_abs_gpu_vs_fp64": dist64(npred, n64), "max_abs_literal_fp32_vs_fp64": dist64(n32, n64),
                             "max_abs_gpu_vs_literal_fp32": [round(float((npred[s] - n32[s]).abs().max()), 6) for s in range(4)]}

        # ... and the pair BASELINE config 1 names (the reference's own KITTI frame, tests/golden/kitti_pair/: image data,
        # cropped to 368x1232 and normalised as inference.py:94-103 does): the same account on real image statistics
        kp = os.path.join(ROOT, "tests", "golden", "kitti_pair")
        if os.path.isfile(os.path.join(kp, "left_test.png")) and tuple(margs.maxdisplist) == (24, 5, 5):
            from lwsnet_amd import imageio
            kl = imageio.to_input(imageio.crop_bottom_right(imageio.load_rgb(os.path.join(kp, "left_test.png"))))[None]
            kr = imageio.to_input(imageio.crop_bottom_right(imageio.load_rgb(os.path.join(kp, "right_test.png"))))[None]
            kpred = [p.cpu() for p in model(kl, kr)]
            k32 = lws_oracle.forward(kl, kr, sd, margs.maxdisplist)
            k64 = lws_oracle.forward(kl, kr, sd, margs.maxdisplist, dtype=torch.float64)
            cpu["reference_pair_368x1232"] = {"max_abs_gpu_vs_fp64": dist64(kpred, k64), "max_abs_literal_fp32_vs_fp64": dist64(k32, k64),
                                              "max_abs_gpu_vs_literal_fp32": [round(float((kpred[s] - k32[s]).abs().max()), 6) for s in range(4)],
                                              "what": "reference/left_test.png + right_test.png (config 1's pair), seeded weights"}
            _lib.check(lib.lws_reserve(model._h, B, H, W), "lws_reserve")

    pairs = world * B * args.steps
    dtype_name = "f32 (fp16-rounded features)" if args.feature_fp16 else "f32"
    if model.get_option("split_bf16") != 0:
        # experiments only (--opt): the opt-in numerics mode, float32-level accuracy but not the oracle's bits
        dtype_name += " with split-bf16 MFMA operands (3 x bf16 per f32, f32 accumulate; not bit-exact against the oracle)"
    roof = _with_traffic(roof, B, H, W, args.maxdisp0, args.feature_fp16)
    measured, why_not = getattr(args, "traffic_in_run", (None, "not attempted (--no-measure-traffic, or not a plain single-GPU run)"))
    if roof is not None and measured is not None:
        # counters read in this run (two rocprofv3 --pmc child passes before the timed run); the committed summary stays beside it
        roof["traffic_committed"] = roof.get("traffic")
        roof["traffic"] = measured["traffic"]
        roof["traffic_source"] = ("this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of the same workload (3 steps, "
                                  f"{measured['launches_counted']} launches of the kernel each), 2 x FETCH + WRITE; {measured['seconds']} s")
        roof["traffic_counters_kb"] = {"FETCH_SIZE": measured["FETCH_SIZE_KB_avg"], "WRITE_SIZE": measured["WRITE_SIZE_KB_avg"]}
        roof.pop("traffic_note", None)
    elif roof is not None:
        roof["traffic_in_run_note"] = why_not
    if roof is not None:
        # `traffic`: counters of this run's own rocprofv3 --pmc child passes (plain single-GPU runs), else the committed summary of
        # this workload (keyed on the kernel source's sha256); under --gpus N > 1 the object describes rank 0's launches
        roof["traffic_measured_in_run"] = measured is not None
        roof["rank"] = 0
    out = {
        "metric": "stereo pairs/sec @256x512 maxdisp=192 (stage-4)",
        "value": None if args.one_gpu else round(pairs / elapsed, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "spinup_s": round(spun_s, 3), "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": (f"BASELINE config 2: batch={B}/GPU, {H}x{W} synthetic pair, maxdisplist=[24,5,5], all 4 stages"
                                if (H, W, args.maxdisp0) == (256, 512, 24) else
                                f"batch={B}/GPU, {H}x{W} synthetic pair, maxdisplist=[{args.maxdisp0},5,5], all 4 stages"),
                   "pairs_per_gpu": B, "streams": S, "parallelism": f"batch-sharded x{world}, 1 RCCL gather of stage-4 per {G * B} pairs per rank" if grouped else "single GPU",
                   "weights": "seeded synthetic (seed 7, calibrated BN)", "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), **({"options": args.opt} if args.opt else {})},
        "roofline": roof, "secondary": secondary, "cpu_baseline": cpu, "latency_ms": latency, "pipelined": pipelined,
        "split_bf16": split_bf16, "config4": config4,
        "hbm_kernels": hbm,
        "hot_path_kernel_ms_per_step": round(hot_ms, 4), "all_kernel_ms_per_step": round(all_ms, 4), "kernels": {k: {a: round(b, 2) for a, b in v.items()} for k, v in kernels.items()},
    }
    if grouped:
        out["collective"] = {"backend": "gloo-through-host" if through_host else dist.get_backend(),
                             "op": f"async gather of stage-4 maps to rank 0, one per {G} step(s) = {G * B} pairs per rank per gather",
                             "gather_every_steps": G, "gathers_in_timed_region": leg["gathers"],
                             "world": world, "rank0_slot_equals_local": gather_ok, "all_ranks_slots_equal_unsharded": all_ok}
        if args.one_gpu:
            out["shared_one_gpu"] = {"pairs_per_s_both_ranks_on_one_gpu": round(pairs / elapsed, 2),
                                     "note": f"{world} ranks of the HIP path sharing cuda:0 over gloo (device maps gathered through the "
                                             "host): a check of the N-rank code path, not a throughput -- value is null"}
        if through_host:
            # (the gloo form copies every shard through the host and blocks the issuing thread: its cost is not the collective's)
            out["collective"]["overhead_pct"] = None
            out["collective"]["note"] = ("device maps gathered over gloo through host memory: every gather synchronises the host, so "
                                         "ms_per_step includes host round trips an RCCL gather does not have")
        if collective_overhead:
            base = collective_overhead["ms_per_step_without_gather"]
            collective_overhead["overhead_pct"] = round(100.0 * (out["ms_per_step"] - base) / base, 2)
            out["collective"].update(collective_overhead)
    print(json.dumps(out), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
