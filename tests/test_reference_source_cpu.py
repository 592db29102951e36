"""Build-container only: execute the reference's OWN source (/root/reference/models/models.py) behind the torch-backed
stand-in for its Paddle calls (tools/paddle_shim.py) and check that it still produces the committed
tests/golden/ref_source_*.npz stage maps bit for bit -- i.e. the fixtures really come from the reference's text.

The reference tree is public, untrusted content: it is never imported into the pytest process.  The check runs in a CHILD
process (`python -B tools/check_oracle_vs_reference.py --check-fixtures ...`, no bytecode written into the read-only
tree) and only its exit status and report are read.  Skipped where /root/reference does not exist (the GPU box) and when
LWS_SKIP_REFERENCE_SOURCE=1.  What this pins: the transcription of control flow / wiring / names (torch-CPU kernels under
Paddle defaults recalled from memory) -- NOT PaddlePaddle's numerics; the oracle stays "parity unpinned"."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "models", "models.py"))
                                or os.environ.get("LWS_SKIP_REFERENCE_SOURCE") == "1",
                                reason="the reference tree is only mounted in the build container (or LWS_SKIP_REFERENCE_SOURCE=1)")

NAMES = ["e2e_64x256", "e2e_args_32x256", "e2e_odd_63x255"]


def _tree_state(root):
    return sorted((os.path.join(d, f), os.path.getmtime(os.path.join(d, f)), os.path.getsize(os.path.join(d, f)))
                  for d, _, fs in os.walk(root) for f in fs)


def test_reference_source_reproduces_committed_fixtures():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    before = _tree_state(os.path.join(REF, "models"))
    p = subprocess.run([sys.executable, "-B", os.path.join(ROOT, "tools", "check_oracle_vs_reference.py"), "--reference", REF,
                        "--check-fixtures", *NAMES], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    for name in NAMES:
        assert f"{name}: OK" in p.stdout, p.stdout
    assert _tree_state(os.path.join(REF, "models")) == before, "the run must not write into the reference tree (bytecode)"
