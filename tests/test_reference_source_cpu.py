"""Build-container only: execute the reference's OWN source (/root/reference/models/models.py, imported in place) behind the
torch-backed stand-in for its Paddle calls (tools/paddle_shim.py) and check that it still produces the committed
tests/golden/ref_source_*.npz stage maps bit for bit -- i.e. the fixtures really come from the reference's text.
Skipped where /root/reference does not exist (the GPU box); nothing from the reference is copied or travels."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "models", "models.py")),
                                reason="the reference tree is only mounted in the build container")


@pytest.mark.parametrize("name", ["e2e_64x256", "e2e_args_32x256", "e2e_odd_63x255"])
def test_reference_source_reproduces_committed_fixture(name):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_oracle_vs_reference as chk
    from lwsnet_amd.weights import default_args, make_state_dict
    g = golden(f"ref_source_{name}.npz")
    args = default_args(maxdisplist=tuple(int(v) for v in g["maxdisplist"]), layers_3d=int(g["layers_3d"]),
                        channels_3d=int(g["channels_3d"]), growth_rate=tuple(int(v) for v in g["growth_rate"]))
    sd = make_state_dict(int(g["seed"]), args, calibrated=bool(g["calibrated"]))
    try:
        out, keys = chk.run_reference(REF, args, sd, g["left"], g["right"], torch.float32)
    finally:
        for m in [k for k in sys.modules if k == "paddle" or k.startswith("paddle.") or k == "models" or k.startswith("models.")]:
            del sys.modules[m]                       # leave no stand-in behind for other tests
        if REF in sys.path:
            sys.path.remove(REF)
    assert keys == sorted(sd.keys())                 # the 226 structured names are the reference's own
    for i in range(4):
        assert np.array_equal(out[i], g[f"pred{i}"]), f"{name} stage {i + 1}"
