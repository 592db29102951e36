"""ctypes binding of liblwsnet_hip.so (include/lwsnet_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblwsnet_hip.so")

LWS_OK, LWS_ERR_INVALID, LWS_ERR_HIP, LWS_ERR_STATE, LWS_ERR_NOMEM = 0, -1, -2, -3, -4

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int64_p = ctypes.POINTER(ctypes.c_int64)


class LwsConfig(ctypes.Structure):
    _fields_ = [("maxdisplist", ctypes.c_int32 * 3), ("layers_3d", ctypes.c_int32),
                ("channels_3d", ctypes.c_int32), ("growth_rate", ctypes.c_int32 * 3),
                ("feature_fp16", ctypes.c_int32), ("interp_align_mode", ctypes.c_int32)]


# name -> (restype, argtypes); exactly the functions include/lwsnet_hip.h declares
_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
PROTOTYPES = {
    "lws_abi_version": (_i, []),
    "lws_last_error": (ctypes.c_char_p, []),
    "lws_device_count": (_i, []),
    "lws_create": (_i, [ctypes.POINTER(LwsConfig), ctypes.POINTER(_vp)]),
    "lws_destroy": (_i, [_vp]),
    "lws_set_tensor": (_i, [_vp, ctypes.c_char_p, c_float_p, c_int64_p, _i]),
    "lws_finalize": (_i, [_vp]),
    "lws_reserve": (_i, [_vp, _i, _i, _i]),
    "lws_volume_l1_shift": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lws_volume_l1_warp": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "lws_conv3d_stack": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "lws_softargmin": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "lws_upsample_add": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lws_disparity_stages": (_i, [_vp, _vp * 3, _vp * 3, _i, _i, _i, _vp * 3, _vp]),
    "lws_feature_extraction": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lws_refine": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "lws_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp * 4, _vp]),
    "lws_preprocess_rgb8": (_i, [_vp, _vp, _i, _i, _i, c_float_p, c_float_p, _vp]),
    "lws_apply_lut8": (_i, [_vp, _vp, _vp, ctypes.c_int64, _vp]),
    "lws_set_option": (_i, [_vp, ctypes.c_char_p, _i]),
    "lws_get_option": (_i, [_vp, ctypes.c_char_p, ctypes.POINTER(_i)]),
    "lws_profile_enable": (_i, [_vp, _i]),
    "lws_profile_sample": (_i, [_vp, _i]),
    "lws_profile_read": (_i, [_vp, ctypes.POINTER(ctypes.c_double), c_int64_p]),
    "lws_profile_read_class": (_i, [_vp, _i, c_float_p, _i, ctypes.POINTER(_i)]),
    "lws_kernel_class_name": (ctypes.c_char_p, [_i]),
    "lws_clock_stamp": (_i, [_vp, _i]),
    "lws_clock_read": (_i, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "lws_clone": (_i, [_vp, ctypes.POINTER(_vp)]),
    "lws_pool_create": (_i, [_vp, _i, _i, ctypes.POINTER(_vp)]),
    "lws_pool_destroy": (_i, [_vp]),
    "lws_pool_workers": (_i, [_vp]),
    "lws_pool_reserve": (_i, [_vp, _i, _i, _i]),
    "lws_pool_submit": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp * 4, _vp, c_int64_p]),
    "lws_pool_wait": (_i, [_vp, ctypes.c_int64]),
    "lws_pool_wait_all": (_i, [_vp]),
    "lws_pool_clear_error": (_i, [_vp]),
    "lws_pool_profile_enable": (_i, [_vp, _i, _i]),
    "lws_pool_profile_read": (_i, [_vp, ctypes.POINTER(ctypes.c_double), c_int64_p]),
}
LWS_POOL_SIDE_STREAMS = 1
LWS_KC_COUNT = 13

_lib = None


def load():
    """Loads the library (building is the job of __graft_entry__.build / lwsnet_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so).  It must be the first HIP runtime in
    # the process: loading this library first pulls /opt/rocm's copy in, and the second runtime then reports "no
    # ROCm-capable device".  With torch imported first, the SONAME libamdhip64.so.7 resolves to torch's copy and both
    # share one runtime (device pointers and streams interoperate).
    import torch  # noqa: F401
    from . import late_env
    if late_env():
        import warnings
        warnings.warn(f"lwsnet_amd was imported after HIP had initialised in this process: {', '.join(late_env())} could not be "
                      "exported in time (import lwsnet_amd before the first torch.cuda call, or export them in the environment); "
                      "multi-stream forwards / RCCL may run slower or fail (lwsnet_amd/__init__.py)", RuntimeWarning, stacklevel=2)
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -m lwsnet_amd.build`); there is no CPU fallback for the disparity path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.lws_abi_version() != 8:
        raise RuntimeError("liblwsnet_hip.so ABI version mismatch; rebuild the extension")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc == LWS_OK:
        return
    msg = load().lws_last_error().decode("utf-8", "replace")
    if rc == LWS_ERR_INVALID:
        raise ValueError(msg or what)
    raise RuntimeError(f"{what}: {msg} (status {rc})")
