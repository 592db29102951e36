"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol
include/lwsnet_hip.h declares; host-side argument / state errors behave like the
reference's Python exceptions (ValueError for bad shapes, RuntimeError otherwise)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from lwsnet_amd import _lib
from lwsnet_amd.weights import default_args, make_state_dict


def _declared():
    txt = open(os.path.join(ROOT, "include", "lwsnet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lws_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_prototypes_agree():
    assert _declared() == sorted(_lib.PROTOTYPES)


def test_library_exports_every_symbol(hip_lib):
    for name in _declared():
        assert hasattr(hip_lib, name), name
    assert hip_lib.lws_abi_version() == 8


def _create(lib, **kw):
    a = default_args(**kw)
    cfg = _lib.LwsConfig((ctypes.c_int32 * 3)(*a.maxdisplist), a.layers_3d, a.channels_3d,
                         (ctypes.c_int32 * 3)(*a.growth_rate), 0, a.interp_align_mode)
    h = ctypes.c_void_p()
    rc = lib.lws_create(ctypes.byref(cfg), ctypes.byref(h))
    return rc, h


def test_create_rejects_unsupported_config(hip_lib):
    rc, _ = _create(hip_lib, channels_3d=5)
    assert rc == _lib.LWS_ERR_INVALID
    with pytest.raises(ValueError, match="not supported"):
        _lib.check(rc)
    rc, _ = _create(hip_lib, maxdisplist=(100, 5, 5))
    assert rc == _lib.LWS_ERR_INVALID
    rc, _ = _create(hip_lib, interp_align_mode=2)          # lws_config.interp_align_mode (ABI v8): 0 or 1
    assert rc == _lib.LWS_ERR_INVALID and b"interp_align_mode" in hip_lib.lws_last_error()
    rc, h = _create(hip_lib, interp_align_mode=1)
    assert rc == 0
    hip_lib.lws_destroy(h)


def test_set_tensor_validates_keys_and_shapes(hip_lib):
    rc, h = _create(hip_lib)
    assert rc == 0
    sd = make_state_dict(7)
    k = "volume_postprocess.0.1.2.weight"
    v = sd[k]
    shp = (ctypes.c_int64 * v.ndim)(*v.shape)
    assert hip_lib.lws_set_tensor(h, k.encode(), v.ctypes.data_as(_lib.c_float_p), shp, v.ndim) == 0
    assert hip_lib.lws_set_tensor(h, b"no.such.key", v.ctypes.data_as(_lib.c_float_p), shp, v.ndim) == _lib.LWS_ERR_INVALID
    assert b"unexpected key" in hip_lib.lws_last_error()
    bad = (ctypes.c_int64 * 2)(3, 3)
    assert hip_lib.lws_set_tensor(h, k.encode(), v.ctypes.data_as(_lib.c_float_p), bad, 2) == _lib.LWS_ERR_INVALID
    assert b"shape mismatch" in hip_lib.lws_last_error()
    # finalize without the full state dict is a state error, not a crash
    assert hip_lib.lws_finalize(h) == _lib.LWS_ERR_STATE
    with pytest.raises(RuntimeError):
        _lib.check(_lib.LWS_ERR_STATE, "lws_finalize")
    assert hip_lib.lws_destroy(h) == 0


def test_model_shim_validates_on_host(hip_lib):
    from lwsnet_amd.models import LWSNet
    m = LWSNet(default_args(), device=None) if False else None   # constructed below without a device
    import torch
    if torch.cuda.is_available():
        pytest.skip("host-only behaviour")
    m = LWSNet(default_args())
    assert m.eval() is m
    with pytest.raises(NotImplementedError):
        m.train()
    with pytest.raises(KeyError):
        m.set_state_dict({"feature_extraction.dres0.0.0.weight": np.zeros((4, 3, 3, 3), np.float32)})
    sd = make_state_dict(7)
    sd["StructuredToParameterName@@"] = {}
    m.set_state_dict(sd)                      # host-side ingest works without a GPU
    assert len(m.state_dict()) == 226
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(np.zeros((1, 3, 64, 256), np.float32), np.zeros((1, 3, 64, 256), np.float32))


def test_check_size_matches_reference_constraints():
    from lwsnet_amd.synth import check_size
    check_size(256, 512)
    check_size(368, 1232)
    check_size(544, 960, 32)
    for H, W in ((375, 1242), (540, 960), (256, 128)):
        with pytest.raises(ValueError):
            check_size(H, W)


def test_options_validate_names_and_ranges(hip_lib):
    """lws_set_option / lws_get_option are host-side: they work without a GPU."""
    rc, h = _create(hip_lib)
    assert rc == 0
    v = ctypes.c_int(123)
    for name, default in [(b"fuse_first", 3), (b"defer_upsample", 1), (b"side_streams", 1), (b"split_bf16", 0), (b"ref_chunk_mb", 72),
                          (b"ref_pipe", -1), (b"warp_form", 1), (b"fuse_last1", 1), (b"fork2_after", -1), (b"fuse_ref_last", -1)]:
        assert hip_lib.lws_get_option(h, name, ctypes.byref(v)) == 0 and v.value == default
    assert hip_lib.lws_set_option(h, b"fork2_after", 2) == 0
    assert hip_lib.lws_get_option(h, b"fork2_after", ctypes.byref(v)) == 0 and v.value == 2
    assert hip_lib.lws_set_option(h, b"fork2_after", 17) == _lib.LWS_ERR_INVALID
    assert hip_lib.lws_set_option(h, b"fuse_first", 4) == _lib.LWS_ERR_INVALID
    assert hip_lib.lws_set_option(h, b"warp_form", 2) == _lib.LWS_ERR_INVALID
    assert hip_lib.lws_set_option(h, b"bogus", 1) == _lib.LWS_ERR_INVALID
    assert b"unknown option" in hip_lib.lws_last_error()
    # ABI v8 retired the options two rounds of sweeps had shown to tie or lose, and the per-kernel forms of the numerics mode
    for name in (b"left_at", b"split_heads", b"fuse_shift", b"conv3d_order", b"mid8_tile", b"mid8_balance", b"fork_ext", b"tail_at",
                 b"mid8_form", b"mid16_form", b"conv64_form"):
        assert hip_lib.lws_set_option(h, name, 0) == _lib.LWS_ERR_INVALID and hip_lib.lws_get_option(h, name, ctypes.byref(v)) == _lib.LWS_ERR_INVALID
    # the opt-in numerics mode: a bit per MFMA convolution that has a split-bf16 form
    for mask in (1, 2, 4, 7, 0):
        assert hip_lib.lws_set_option(h, b"split_bf16", mask) == 0
        assert hip_lib.lws_get_option(h, b"split_bf16", ctypes.byref(v)) == 0 and v.value == mask
    assert hip_lib.lws_set_option(h, b"split_bf16", 8) == _lib.LWS_ERR_INVALID
    hip_lib.lws_destroy(h)


def test_handle_refuses_a_foreign_current_device(hip_lib):
    """include/lwsnet_hip.h: a handle belongs to one HIP device and every GPU-touching call checks that it is the calling
    thread's current device (a launch from another device would run on foreign pointers).  Host-side check: it fires
    before any HIP work, so it is testable without a GPU -- rebind the handle to device 5 (legal before anything is
    allocated) and call in from whatever device is current here (-1 on a CPU box, 0 on the one-GPU box)."""
    rc, h = _create(hip_lib)
    assert rc == 0
    v = ctypes.c_int(99)
    assert hip_lib.lws_get_option(h, b"device", ctypes.byref(v)) == 0 and v.value in (-1, 0)
    assert hip_lib.lws_set_option(h, b"device", -3) == _lib.LWS_ERR_INVALID
    assert hip_lib.lws_set_option(h, b"device", 5) == 0
    assert hip_lib.lws_reserve(h, 1, 64, 256) == _lib.LWS_ERR_INVALID
    msg = hip_lib.lws_last_error()
    assert b"belongs to HIP device 5" in msg and b"current device" in msg
    with pytest.raises(ValueError, match="belongs to HIP device 5"):
        _lib.check(_lib.LWS_ERR_INVALID)
    z = np.zeros((1, 3, 64, 256), np.float32)
    outs = (ctypes.c_void_p * 4)(*[z.ctypes.data] * 4)
    assert hip_lib.lws_forward(h, z.ctypes.data, z.ctypes.data, 1, 64, 256, outs, None) == _lib.LWS_ERR_INVALID
    assert b"lws_forward: the handle belongs to HIP device 5" in hip_lib.lws_last_error()
    assert hip_lib.lws_finalize(h) == _lib.LWS_ERR_INVALID          # device check comes before the state check
    # size errors still win over the device error (pure argument validation comes first)
    assert hip_lib.lws_reserve(h, 1, 375, 1242) == _lib.LWS_ERR_INVALID
    assert b"unsupported input size" in hip_lib.lws_last_error()
    assert hip_lib.lws_destroy(h) == 0


def test_clone_and_pool_validate_on_host(hip_lib):
    """lws_clone / lws_pool_* argument and state errors need no GPU."""
    rc, h = _create(hip_lib)
    assert rc == 0
    c = ctypes.c_void_p()
    assert hip_lib.lws_clone(h, ctypes.byref(c)) == _lib.LWS_ERR_STATE         # not finalized
    assert b"not been finalized" in hip_lib.lws_last_error()
    p = ctypes.c_void_p()
    assert hip_lib.lws_pool_create(h, 0, 0, ctypes.byref(p)) == _lib.LWS_ERR_INVALID
    assert hip_lib.lws_pool_create(h, 3, 8, ctypes.byref(p)) == _lib.LWS_ERR_INVALID
    assert hip_lib.lws_pool_create(h, 3, 0, ctypes.byref(p)) == _lib.LWS_ERR_STATE
    assert hip_lib.lws_pool_workers(None) == 0 and hip_lib.lws_pool_destroy(None) == 0
    assert hip_lib.lws_pool_wait_all(None) == _lib.LWS_ERR_INVALID
    assert hip_lib.lws_pool_clear_error(None) == _lib.LWS_ERR_INVALID
    assert hip_lib.lws_destroy(h) == 0
