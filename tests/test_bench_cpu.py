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
