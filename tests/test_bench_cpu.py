"""bench.py plumbing without a GPU: `python bench.py --gpus 2` must start the 2-rank job itself (child
`python -m torch.distributed.run`), shard the seeded pairs, gather the stage-4 maps on rank 0 and print
ONE JSON line.  --dry-run-cpu swaps the HIP forward for a per-pair stand-in over gloo; nothing is measured (value null)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _env(**more):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("LWS_BENCH_")}
    env.update(more)
    return env


def _run(*extra, steps="3", env=None, rc=0, dry=True, launcher=()):
    cmd = [sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *(["--dry-run-cpu"] if dry else []), "--steps", steps, "--warmup", "1", *extra]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env or _env(), cwd=ROOT)
    assert (p.returncode == 0) == (rc == 0), (p.returncode, p.stderr[-2000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                    # exactly ONE JSON line, whatever happened
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    d = _run("--gpus", "2", "--batch", "3")
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None
    assert d["gather_equals_unsharded"] is True and d["pairs_gathered"] == 6 and d["gathers"] == 2     # 4 steps, 2 per gather


def test_bench_staged_gather_flushes_the_tail():
    """5 steps with 2 steps per gather: two full staging buffers (alternating) and a tail of one step, flushed before the
    clock stops; the last gather must hold every rank's map of the LAST step (lwsnet_amd.dist.StagedGather)."""
    d = _run("--gpus", "2", "--batch", "2", steps="4")
    assert d["gather_equals_unsharded"] is True and d["gathers"] == 3 and d["pairs_gathered"] == 4


def test_bench_single_rank_dry_run():
    d = _run("--gpus", "1")
    assert d["n_gpus"] == 1 and d["gather_equals_unsharded"] is True and d["pairs_gathered"] == 1


def test_bench_eight_ranks_ragged_tail():
    """BASELINE config 4's launch shape on CPU: `python bench.py --gpus 8` starts eight gloo ranks, every rank stages its
    pairs, the staged gather runs full buffers plus a ragged tail (5 steps, 2 steps per gather -> 3 gathers, the last with one
    step), rank 0 receives every rank's shard of the last step and prints ONE JSON line."""
    d = _run("--gpus", "8", "--batch", "2", steps="4")
    assert d["n_gpus"] == 8 and d["dry_run"] is True and d["value"] is None
    assert d["gather_equals_unsharded"] is True and d["pairs_gathered"] == 16 and d["gathers"] == 3


def test_bench_n_rank_job_reports_config4():
    """under --gpus N > 1 the job also runs BASELINE config 4's per-rank shape (8 pairs per GPU per step,
    batch 8 N) with its own gather check; `value` stays the --batch leg.  `--gpus 8 --dry-run-cpu` prints `config4`."""
    d = _run("--gpus", "8")
    c4 = d["config4"]
    assert c4["pairs_per_gpu"] == 8 and c4["global_batch"] == 64 and c4["all_ranks_slots_equal_unsharded"] is True
    assert c4["pairs_gathered"] == 64 and d["pairs_gathered"] == 8          # (the main leg: 1 pair per rank)
    assert d["attempts"] == [{"collective": "gloo", "ok": True, "s": d["attempts"][0]["s"]}]
    assert "config4" not in _run("--gpus", "1")


def test_bench_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`: every torchrun rank is a supervisor that
    starts its worker as a fresh child (lwsnet_amd/launch.py); rank 0 relays the one line."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    d = _run("--gpus", "2", "--batch", "2", launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                                                      "127.0.0.1", "--master-port", str(port)))
    assert d["n_gpus"] == 2 and d["gather_equals_unsharded"] is True and d["config4"]["all_ranks_slots_equal_unsharded"] is True
    assert [a["ok"] for a in d["attempts"]] == [True]


def test_bench_init_failure_takes_the_labelled_fallback():
    """A rank that fails at the rendezvous ends every rank's first attempt at once (failure flag in the supervisors' store);
    ONE fallback job runs in fresh workers with the gather over gloo through the host and says so in collective.backend."""
    d = _run("--gpus", "2", "--job-timeout", "60", "--init-timeout", "20", env=_env(LWS_BENCH_INJECT="init-fail:1"))
    assert [a["ok"] for a in d["attempts"]] == [False, True] and d["attempts"][1]["collective"] == "gloo-host"
    assert "injected init failure on rank 1" in d["attempts"][0]["reason"]
    assert d["collective"]["backend"].startswith("gloo-through-host (") and "job failed" in d["collective"]["backend"]
    assert d["gather_equals_unsharded"] is True and d["attempts"][0]["s"] < 30


def test_bench_hung_rank_is_killed_by_the_watchdog():
    """A rank that hangs before the rendezvous: the others' init_process_group times out (--init-timeout) or, failing that, the
    supervisors' deadline (--job-timeout) kills the workers' process groups; the fallback then measures."""
    d = _run("--gpus", "2", "--job-timeout", "12", "--init-timeout", "60", env=_env(LWS_BENCH_INJECT="hang:1"))
    assert [a["ok"] for a in d["attempts"]] == [False, True]
    assert "job timeout" in d["attempts"][0]["reason"] and 12 <= d["attempts"][0]["s"] < 25
    assert d["collective"]["backend"].startswith("gloo-through-host (")


def test_bench_hang_in_the_first_collective():
    """A rank that never enters the first gather: the process group's timeout (--init-timeout applies to every collective)
    fails the waiting ranks, the fallback takes over."""
    d = _run("--gpus", "2", "--job-timeout", "60", "--init-timeout", "8", env=_env(LWS_BENCH_INJECT="hang-collective:1"))
    assert [a["ok"] for a in d["attempts"]] == [False, True] and d["attempts"][0]["s"] < 40


def test_bench_both_attempts_fail_prints_the_error_line():
    """Nothing measurable: ONE JSON line with value null and the reasons, exit status != 0 -- never a hang, never silence."""
    d = _run("--gpus", "2", "--job-timeout", "6", env=_env(LWS_BENCH_INJECT="hang:0", LWS_BENCH_INJECT_ATTEMPTS="all"), rc=1)
    assert d["value"] is None and "job timeout" in d["error"] and [a["ok"] for a in d["attempts"]] == [False, False]


def test_bench_too_few_devices_fails_fast():
    """The measured (non-dry) job on a node with fewer devices than ranks: the supervisors see it with device_count (which does
    not initialise HIP) and print the error line at once.  (This container has no GPU; on a GPU box the test needs < 2.)"""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a node with fewer than 2 devices")
    d = _run("--gpus", "2", dry=False, rc=4)
    assert d["value"] is None and "HIP device(s) visible" in d["error"] and d["n_gpus"] == 2


def test_gather_policy_is_a_function_of_world_size():
    """lwsnet_amd.dist.gather_policy / gather_every (the root-rank budget of round 4): no channel cap at any world size, 8 pairs
    per rank per gather up to four ranks, 16 on eight; bench.py's cadence follows (--gather-pairs overrides)."""
    from lwsnet_amd import dist as ldist
    assert [ldist.gather_policy(w) for w in (1, 2, 3, 4, 7, 8, 16)] == [(None, 8)] * 5 + [(None, 16)] * 2
    assert ldist.gather_every(8, 8) == 2 and ldist.gather_every(8, 1) == 16 and ldist.gather_every(8, 64) == 1
    assert ldist.gather_every(4, 8) == 1 and ldist.gather_every(2, 1) == 8 and ldist.gather_every(8, 1, min_pairs=4) == 4
    import os
    saved = {k: os.environ.pop(k, None) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MAX_P2P_NCHANNELS")}
    try:
        assert ldist.apply_channel_cap(8) is None and "NCCL_MAX_NCHANNELS" not in os.environ
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


def test_traffic_measurement_falls_back_without_a_gpu():
    """bench.py measures roofline.traffic in the run with two `rocprofv3 --pmc` child passes (plain single-GPU runs).  Whatever
    goes wrong there -- no profiler, no GPU (this container), a refusal, a timeout -- must only cost the in-run number: the
    function returns (None, reason) and never raises, and bench.py then reports the committed summary."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = types.SimpleNamespace(batch=1, size="256x512", maxdisp0=24, feature_fp16=False, opt=[], traffic_timeout=120.0)
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the fallback is what this test is about: needs a box without a GPU")
    measured, why = bench._measure_traffic(args)
    assert measured is None and isinstance(why, str) and why


def test_launch_child_deadline_and_peer_failure():
    """lwsnet_amd.launch.Child: a worker in its own process group is ended by the deadline, or as soon as a peer has failed,
    and a worker that exits by itself reports its status and its JSON line."""
    import time
    from lwsnet_amd import launch
    sleeper = [sys.executable, "-c", "import json, time; print(json.dumps({'started': True}), flush=True); time.sleep(120)"]
    t0 = time.monotonic()
    c = launch.Child(sleeper, dict(os.environ))
    ok, why = c.watch(time.monotonic() + 1.0)
    assert not ok and "job timeout" in why and time.monotonic() - t0 < 15 and c.p.poll() is not None
    assert c.json_lines() == ['{"started": true}']
    c = launch.Child(sleeper, dict(os.environ))
    ok, why = c.watch(time.monotonic() + 60.0, peer_failed=lambda: True)
    assert not ok and why == "ended because another rank failed" and c.p.poll() is not None
    c = launch.Child([sys.executable, "-c", "import json; print(json.dumps({'value': 1})); raise SystemExit(0)"], dict(os.environ))
    assert c.watch(time.monotonic() + 30.0) == (True, None) and c.json_lines() == ['{"value": 1}']
    c = launch.Child([sys.executable, "-c", "raise SystemExit(7)"], dict(os.environ))
    assert c.watch(time.monotonic() + 30.0) == (False, "worker exited with status 7")
    d = json.loads(launch.error_line(8, 20, 5, "why"))
    assert d["value"] is None and d["n_gpus"] == 8 and d["error"] == "why" and d["unit"] == "pairs/s"
