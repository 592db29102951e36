"""Checkpoint ingest for the reference's `.pdparams` files.

`paddle.save(model.state_dict(), path)` in Paddle 2.0.0rc0 (/root/reference/train.py:115) writes a pickle of
`{structured name: ndarray}` plus a `"StructuredToParameterName@@"` bookkeeping dict; Paddle >= 2.1 stores each
tensor as a `(name, ndarray)` tuple.  No sample file ships with the reference (README.md:122-123 are Drive links), so
the format knowledge is from the Paddle sources; the loader is a RESTRICTED unpickler (numpy reconstruction only -- a
checkpoint can never execute code) and also accepts `.npz`.
"""
from __future__ import annotations

import collections
import io
import pickle

import numpy as np

_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("collections", "OrderedDict"),
    ("_codecs", "encode"),                       # protocol-2 pickles of ndarrays written by Python 3
}


class _NumpyOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) not in _ALLOWED:
            raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}: only numpy arrays are accepted")
        if module.startswith("numpy.core") or module.startswith("numpy._core"):
            import numpy.core.multiarray as ma
            return getattr(ma, name)
        return super().find_class(module, name)


def load_state_dict(path):
    """Returns {structured name: float32 ndarray} from a .pdparams pickle or an .npz archive."""
    if str(path).endswith(".npz"):
        with np.load(path) as z:
            return {k: np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}
    with open(path, "rb") as f:
        obj = _NumpyOnlyUnpickler(io.BytesIO(f.read())).load()
    if not isinstance(obj, (dict, collections.OrderedDict)):
        raise ValueError(f"{path}: expected a dict of arrays, got {type(obj).__name__}")
    return _state_from_saved(obj, path)


def _state_from_saved(obj, path="<checkpoint>"):
    """The dict `paddle.save` pickled -> {structured name: float32 ndarray}.

    2.0.0rc0 (`_build_saved_state_dict`): `{structured name: ndarray}` plus `"StructuredToParameterName@@"`, a dict from
    structured names to Paddle's internal parameter names (`conv2d_0.w_0`, `batch_norm_3.w_1`, ...); 2.0 final may add
    `"UnpackBigParamInfor@@"` (arrays above 1 GB split into slices: none here, so its presence with entries is refused).
    >= 2.1 (`reduce_varbase`): every tensor as a `(parameter name, ndarray)` tuple, no name table."""
    out = {}
    for k, v in obj.items():
        if not isinstance(k, str):
            raise ValueError(f"{path}: key {k!r} is not a string")
        if k.endswith("@@"):
            if k == "UnpackBigParamInfor@@" and v:
                raise ValueError(f"{path}: the checkpoint holds arrays split by paddle.save ({sorted(v)}): not supported")
            if not isinstance(v, (dict, collections.OrderedDict)):
                raise ValueError(f"{path}: bookkeeping entry '{k}' is {type(v).__name__}, not a dict")
            continue
        if isinstance(v, (tuple, list)) and len(v) == 2 and isinstance(v[0], str) and isinstance(v[1], np.ndarray):
            v = v[1]
        if not isinstance(v, np.ndarray):
            raise ValueError(f"{path}: entry '{k}' is {type(v).__name__}, not an ndarray")
        if v.dtype.kind != "f":
            raise ValueError(f"{path}: entry '{k}' has dtype {v.dtype}, expected a floating-point array")
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def save_pdparams(state_dict, path):
    """Writes the 2.0rc0 layout (protocol 2 pickle) -- used by the tests and to hand weights back to the reference."""
    obj = {k: np.asarray(v) for k, v in state_dict.items()}
    obj["StructuredToParameterName@@"] = {k: k for k in state_dict}
    with open(path, "wb") as f:
        pickle.dump(obj, f, protocol=2)


def save_npz(state_dict, path):
    np.savez_compressed(path, **{k: np.asarray(v, dtype=np.float32) for k, v in state_dict.items()})
