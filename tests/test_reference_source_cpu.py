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

NAMES = ["e2e_64x256", "e2e_args_32x256", "e2e_odd_63x255", "e2e_align1_64x256"]


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


_FAKE_PADDLE = '''"""A FAKE `paddle` package for tests/test_reference_source_cpu.py: the "a real paddle is importable" branch of
tools/check_oracle_vs_reference.py --real-paddle has never met a real PaddlePaddle (none can be installed here), so the test
gives it an importable package that is NOT marked as the stand-in.  Its arithmetic is the stand-in's (tools/paddle_shim.py), so
the expected report is bit-equality with the committed fixtures; what is under test is the mode's control flow: the stand-in
is not installed, the reference's source runs on whatever `import paddle` finds, the gate is evaluated, and a checkpoint
written by `paddle.save` goes through lwsnet_amd.checkpoint."""
import pickle
import sys
import types

import numpy as np
import paddle_shim as _b          # tools/ is on sys.path in the child

__version__ = "0.0-fake-for-tests"
for _n in ("arange", "expand", "reshape", "concat", "transpose", "zeros", "norm", "unsqueeze", "squeeze", "sum", "no_grad",
           "to_tensor", "Tensor"):
    globals()[_n] = getattr(_b, _n)
DEVICE = []


def set_device(name):
    DEVICE.append(name)


def save(state_dict, path):
    """paddle.save of 2.0.0rc0 (train.py:115): {structured name: ndarray} + the name table, pickle protocol 2."""
    obj = {k: np.asarray(v.detach().numpy()) for k, v in state_dict.items()}
    obj["StructuredToParameterName@@"] = {k: "param_%d" % i for i, k in enumerate(state_dict)}
    with open(path, "wb") as f:
        pickle.dump(obj, f, protocol=2)


nn = types.ModuleType("paddle.nn")
for _n in ("Layer", "Sequential", "LayerList", "ReLU", "Conv2D", "Conv3D", "Conv2DTranspose", "BatchNorm2D", "BatchNorm3D"):
    setattr(nn, _n, getattr(_b, _n))
nn.initializer = types.ModuleType("paddle.nn.initializer")
nn.initializer.KaimingNormal = _b.KaimingNormal
nn.functional = types.ModuleType("paddle.nn.functional")
for _n in ("relu", "softmax", "interpolate", "grid_sample"):
    setattr(nn.functional, _n, getattr(_b, _n))
import os as _os
if _os.environ.get("LWS_FAKE_PADDLE_ALIGN_MODE") == "1":        # a Paddle whose F.interpolate defaults to align_mode = 1
    nn.functional.interpolate = lambda x, size=None, mode="nearest", align_corners=False: _b.interpolate(x, size, mode, align_corners, align_mode=1)
sys.modules.update({"paddle.nn": nn, "paddle.nn.initializer": nn.initializer, "paddle.nn.functional": nn.functional})
'''


def test_real_paddle_mode_with_a_fake_paddle_package(tmp_path):
    """`--real-paddle` must use a real `paddle` when one is importable (no stand-in installed), report the
    per-stage distance to the committed fixtures against the noise-floor gate, and read one Paddle-written `.pdparams` through
    `checkpoint.load_state_dict`.  No Paddle exists here, so the importable package is a fake (see _FAKE_PADDLE); without it
    the mode must say so and exit 2."""
    tool = os.path.join(ROOT, "tools", "check_oracle_vs_reference.py")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    before = _tree_state(os.path.join(REF, "models"))
    p = subprocess.run([sys.executable, "-B", tool, "--reference", REF, "--real-paddle"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert p.returncode == 2 and "nothing checked" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    pkg = tmp_path / "site" / "paddle"
    pkg.mkdir(parents=True)
    (pkg / "__init__.py").write_text(_FAKE_PADDLE)
    kept = tmp_path / "written_by_paddle_save.pdparams"
    env["PYTHONPATH"] = os.pathsep.join([str(tmp_path / "site"), env.get("PYTHONPATH", "")])
    p = subprocess.run([sys.executable, "-B", tool, "--reference", REF, "--real-paddle", "--keep-pdparams", str(kept)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "0.0-fake-for-tests" in p.stdout and "stand-in NOT installed" in p.stdout
    assert p.stdout.count("INSIDE the gate") == 5 and p.stdout.count("(bit-equal)") == 5, p.stdout
    assert "all equal to what was set" in p.stdout and "parity with PaddlePaddle holds" in p.stdout
    assert "follows align_mode = 0 (half-pixel centres) on every case" in p.stdout
    # a Paddle that resizes with src = ratio * dst (the one bet of the oracle that would move every stage): the mode names the
    # reading it matches and what to flip, exit status 3
    env["LWS_FAKE_PADDLE_ALIGN_MODE"] = "1"
    p = subprocess.run([sys.executable, "-B", tool, "--reference", REF, "--real-paddle"], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert p.returncode == 3, (p.stdout[-3000:], p.stderr[-3000:])
    assert "follows align_mode = 1 (src = ratio * dst) on every case" in p.stdout and "interp_align_mode = 1" in p.stdout
    assert p.stdout.count("align_mode 1: ") == 5 and p.stdout.count("-> INSIDE") == 5 and p.stdout.count("OUTSIDE the gate") == 5
    # the kept file is what the product's loader reads for inference.py:45
    from lwsnet_amd import checkpoint
    from lwsnet_amd.weights import default_args, make_state_dict
    got = checkpoint.load_state_dict(str(kept))
    want = make_state_dict(7, default_args())
    assert sorted(got) == sorted(want) and len(got) == 226
    assert _tree_state(os.path.join(REF, "models")) == before
