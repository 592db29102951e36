"""bench.py plumbing without a GPU: `python bench.py --gpus 2` must start the 2-rank job itself (child
`python -m torch.distributed.run`), shard the seeded pairs, gather the stage-4 maps on rank 0 and print
ONE JSON line.  --dry-run-cpu swaps the HIP forward for a per-pair stand-in over gloo; nothing is measured (value null)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*extra, steps="3"):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu", "--steps", steps, "--warmup", "1", *extra],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
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
