import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def state_dict():
    from lwsnet_amd.weights import make_state_dict
    return make_state_dict(7)


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library; building is allowed in tests (hipcc cross-compiles on CPU)."""
    from lwsnet_amd import _lib, build
    build.build_library()
    return _lib.load()
