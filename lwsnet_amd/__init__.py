"""lwsnet_amd -- the MI355X-native LWSNet disparity path (HIP kernels behind a C ABI).

Importing the package (any `import lwsnet_amd.<module>`) exports the two runtime switches every entry point of this build
needs BEFORE HIP initialises -- the ROCm runtime reads them once, when the first HIP call creates its queues:

* GPU_MAX_HW_QUEUES=8 -- ROCm maps HIP streams onto hardware queues round-robin (default 4); a forward uses 2 streams, PyTorch,
  RCCL and the lws_pool workers add theirs, and two streams on one queue serialise: under `torchrun` the default cost 12 % of
  every step, the 4-worker pool 13 % (profiles/NOTES.md, round 3);
* HSA_ENABLE_IPC_MODE_LEGACY=0 -- the driver of this pool only supports dmabuf IPC; without it RCCL across processes fails with
  `hipIpcGetMemHandle: invalid argument`.

Values the caller exported are kept (`setdefault`).  If HIP was already up when the package was first imported the switches
cannot take effect any more: `late_env()` names them and `_lib.load()` warns once (bench.py used to be the
only entry point that set them).
"""
import os
import sys

ENV_DEFAULTS = {"GPU_MAX_HW_QUEUES": "8", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
_LATE = []


def _hip_is_up():
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


def _apply_env_defaults():
    up = _hip_is_up()
    for k, v in ENV_DEFAULTS.items():
        if k not in os.environ:
            os.environ[k] = v
            if up:
                _LATE.append(k)


def late_env():
    """The switches this import exported AFTER HIP had initialised in this process (i.e. without effect on it)."""
    return list(_LATE)


_apply_env_defaults()
