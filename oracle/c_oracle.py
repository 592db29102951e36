"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/lws_oracle.c (numpy in, numpy out).

Also holds ``conv3d_stack`` / ``disparity_stages``: the hot path composed from the
C functions exactly as /root/reference/models/models.py:115-156 composes it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblws_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    src = os.path.join(_HERE, "lws_oracle.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def expf(x):
    x = _c(x)
    y = np.empty_like(x)
    lib().lwso_expf_array(_p(x), _p(y), ctypes.c_int64(x.size))
    return y


def volume_l1_shift(L, R, D):
    L, R = _c(L), _c(R)
    B, C, h, w = L.shape
    cost = np.empty((B, D, h, w), np.float32)
    lib().lwso_volume_l1_shift(_p(L), _p(R), _p(cost), B, C, h, w, D)
    return cost


def resize_bilinear(x, hout, wout, mul_a=1.0, mul_b=1.0):
    x = _c(x)
    lead = x.shape[:-2]
    hin, win = x.shape[-2:]
    n = int(np.prod(lead)) if lead else 1
    out = np.empty(lead + (hout, wout), np.float32)
    lib().lwso_resize_bilinear(_p(x), _p(out), n, hin, win, hout, wout,
                               ctypes.c_float(mul_a), ctypes.c_float(mul_b))
    return out


def volume_l1_warp(L, R, wflow, m):
    L, R, wflow = _c(L), _c(R), _c(wflow)
    B, C, h, w = L.shape
    assert wflow.size == B * h * w
    cost = np.empty((B, 2 * m - 1, h, w), np.float32)
    lib().lwso_volume_l1_warp(_p(L), _p(R), _p(wflow), _p(cost), B, C, h, w, m)
    return cost


def bnrelu_conv3d(x, wgt, bn_s, bn_t, skip=None):
    x, wgt, bn_s, bn_t = _c(x), _c(wgt), _c(bn_s), _c(bn_t)
    B, Cin, D, h, w = x.shape
    Cout = wgt.shape[0]
    assert wgt.shape == (Cout, Cin, 3, 3, 3)
    if skip is not None:
        skip = _c(skip)
        assert Cout == 1 and skip.size == B * D * h * w
    y = np.empty((B, Cout, D, h, w), np.float32)
    lib().lwso_bnrelu_conv3d(_p(x), _p(wgt), _p(bn_s), _p(bn_t), _p(skip), _p(y), B, Cin, Cout, D, h, w)
    return y


def softargmin(cost, start):
    cost = _c(cost)
    B, D, h, w = cost.shape
    assert D <= 64
    out = np.empty((B, h, w), np.float32)
    lib().lwso_softargmin(_p(cost), _p(out), B, D, h, w, ctypes.c_float(start))
    return out


def upsample_add(low, prev, H, W):
    low = _c(low)
    B, h, w = low.shape
    prev = _c(prev) if prev is not None else None
    out = np.empty((B, 1, H, W), np.float32)
    mul_b = np.float32(1.0) / np.float32(h)
    lib().lwso_upsample_add(_p(low), _p(prev), _p(out), B, h, w, H, W,
                            ctypes.c_float(float(H)), ctypes.c_float(float(mul_b)))
    return out


def conv3d_stack(cost, sd, stage):
    """cost [B,D,h,w] -> net(cost) + cost  (models.py:136-138)."""
    from lwsnet_amd.weights import bn_scale_shift
    y = _c(cost)[:, None]
    j = 0
    while f"volume_postprocess.{stage}.{j}.2.weight" in sd:
        p = f"volume_postprocess.{stage}.{j}"
        s, t = bn_scale_shift(sd, p + ".0")
        last = f"volume_postprocess.{stage}.{j + 1}.2.weight" not in sd
        y = bnrelu_conv3d(y, sd[p + ".2.weight"], s, t, skip=cost if last else None)
        j += 1
    return y[:, 0]


def disparity_stages(feats_l, feats_r, H, W, sd, maxdisplist=(24, 5, 5), return_costs=False):
    """The three volume stages; returns [pred1, pred2, pred3] as [B,1,H,W] float32."""
    pred, costs = [], []
    for s in range(3):
        fl, fr = _c(feats_l[s]), _c(feats_r[s])
        h, w = fl.shape[2:]
        if s == 0:
            raw = volume_l1_shift(fl, fr, maxdisplist[0])
            start = 0.0
        else:
            mul_b = np.float32(1.0) / np.float32(H)
            wflow = resize_bilinear(pred[s - 1][:, 0], h, w, float(h), float(mul_b))
            raw = volume_l1_warp(fl, fr, wflow, maxdisplist[s])
            start = float(-maxdisplist[s] + 1)
        cost = conv3d_stack(raw, sd, s)
        if return_costs:
            costs.append((raw, cost))
        low = softargmin(cost, start)
        pred.append(upsample_add(low, pred[s - 1] if s else None, H, W))
    return (pred, costs) if return_costs else pred
