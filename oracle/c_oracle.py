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


def bn_scale_shift(sd, prefix, eps=1e-5):
    """Eval-mode BatchNorm (submodules.py:18,196,226) as ``y = fmaf(x, s, t)`` with float32 ``s, t``:
    s = gamma / sqrt(var + eps), t = beta - mean*s, every step rounded to float32 (the HIP library's host code
    performs the same sequence in C++; the oracle keeps its own copy so that it does not import the product)."""
    g = np.asarray(sd[prefix + ".weight"], dtype=np.float32)
    b = np.asarray(sd[prefix + ".bias"], dtype=np.float32)
    m = np.asarray(sd[prefix + "._mean"], dtype=np.float32)
    v = np.asarray(sd[prefix + "._variance"], dtype=np.float32)
    s = (g / np.sqrt(v + np.float32(eps), dtype=np.float32)).astype(np.float32)
    t = (b - (m * s).astype(np.float32)).astype(np.float32)
    return s, t


def _p(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def expf(x):
    x = _c(x)
    y = np.empty_like(x)
    lib().lwso_expf_array(_p(x), _p(y), ctypes.c_int64(x.size))
    return y


def round_fp16(x):
    x = _c(x)
    y = np.empty_like(x)
    lib().lwso_round_fp16(_p(x), _p(y), ctypes.c_int64(x.size))
    return y


def volume_l1_shift(L, R, D):
    L, R = _c(L), _c(R)
    B, C, h, w = L.shape
    cost = np.empty((B, D, h, w), np.float32)
    lib().lwso_volume_l1_shift(_p(L), _p(R), _p(cost), B, C, h, w, D)
    return cost


def _sync_align_mode():
    """The C restatement's resizes follow the SAME switch as the literal restatement's: oracle.lws_oracle.VARIANT["align_mode"]
    (0 = half-pixel centres, the default every golden fixture was made with; 1 = src = ratio * dst)."""
    from . import lws_oracle
    mode = int(lws_oracle.VARIANT["align_mode"])
    lib().lwso_set_align_mode(mode)
    return mode


def resize_bilinear(x, hout, wout, mul_a=1.0, mul_b=1.0):
    _sync_align_mode()
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
    _sync_align_mode()
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
    y = _c(cost)[:, None]
    j = 0
    while f"volume_postprocess.{stage}.{j}.2.weight" in sd:
        p = f"volume_postprocess.{stage}.{j}"
        s, t = bn_scale_shift(sd, p + ".0")
        last = f"volume_postprocess.{stage}.{j + 1}.2.weight" not in sd
        y = bnrelu_conv3d(y, sd[p + ".2.weight"], s, t, skip=cost if last else None)
        j += 1
    return y[:, 0]


def disparity_stages(feats_l, feats_r, H, W, sd, maxdisplist=(24, 5, 5), return_costs=False, feature_fp16=False):
    """The three volume stages; returns [pred1, pred2, pred3] as [B,1,H,W] float32.
    feature_fp16 (BASELINE config 5): the feature maps are rounded to fp16 before the volumes are built."""
    pred, costs = [], []
    for s in range(3):
        fl, fr = _c(feats_l[s]), _c(feats_r[s])
        if feature_fp16:
            fl, fr = round_fp16(fl), round_fp16(fr)
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


# ---------------------------------------------------------------------------------------------
# 2D networks (feature extractor, refinement) composed from the C functions
# ---------------------------------------------------------------------------------------------
def conv2d(x, w, stride=1, pad=1, dil=1, pre=None, depthwise=False):
    x, w = _c(x), _c(w)
    B, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y = np.empty((B, Cout, Ho, Wo), np.float32)
    ps, pt = (_c(pre[0]), _c(pre[1])) if pre is not None else (None, None)
    lib().lwso_conv2d(_p(x), _p(w), _p(ps), _p(pt), _p(y), B, Cin, Cout, H, W, Ho, Wo, k, stride, pad, dil,
                      1 if depthwise else 0)
    return y


def deconv2d_s2(x, w):
    x, w = _c(x), _c(w)
    B, Cin, H, W = x.shape
    Cout = w.shape[1]
    y = np.empty((B, Cout, 2 * H, 2 * W), np.float32)
    lib().lwso_deconv2d_s2(_p(x), _p(w), _p(y), B, Cin, Cout, H, W)
    return y


def bn_add_relu(x, st=None, add=None, relu=False):
    x = _c(x)
    B, C = x.shape[:2]
    plane = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    s, t = (_c(st[0]), _c(st[1])) if st is not None else (None, None)
    lib().lwso_bn_add_relu(_p(x), _p(s), _p(t), _p(_c(add)) if add is not None else None, _p(y), B, C,
                           ctypes.c_int64(plane), 1 if relu else 0)
    return y


def feature_extraction(x, sd):
    """submodules.py:176-188 with convbn's padding rule (:14): [1/8 (16 ch), 1/4 (16 ch), 1/2 (8 ch)]."""
    bn = bn_scale_shift
    fe = "feature_extraction"

    def convbn(v, name, stride, pad, dil, add=None, relu=True):
        y = conv2d(v, sd[f"{fe}.{name}.0.weight"], stride, dil if dil > 1 else pad, dil)
        return bn_add_relu(y, bn(sd, f"{fe}.{name}.1"), add, relu)

    def deconvbn(v, name, add, relu):
        y = deconv2d_s2(v, sd[f"{fe}.{name}.0.weight"])
        return bn_add_relu(y, bn(sd, f"{fe}.{name}.1"), add, relu)

    o = convbn(x, "dres0.0", 2, 1, 2)
    o = convbn(o, "dres0.2", 1, 1, 4)
    r = convbn(o, "dres1.0", 1, 1, 2)
    o = convbn(r, "dres1.2", 1, 1, 2, add=o, relu=False)
    c1 = convbn(o, "dres2.conv1.0", 2, 1, 1)
    pre = convbn(c1, "dres2.conv2.0", 1, 1, 1)
    c3 = convbn(pre, "dres2.conv3.0", 2, 1, 1)
    f8 = convbn(c3, "dres2.conv4.0", 1, 1, 1)
    f4 = deconvbn(f8, "dres2.conv5", pre, True)
    o = deconvbn(f4, "dres2.conv6", o, False)
    o = convbn(o, "classif1.0", 1, 1, 1)
    f2 = conv2d(o, sd[f"{fe}.classif1.2.weight"], 1, 1, 1)
    return [f8, f4, f2]


def refine(left, pred3, sd):
    """models.py:158-162 + submodules.py:223-327: returns pred4 [B,1,H,W]."""
    bn = bn_scale_shift

    def dws(v, prefix, dil):
        v = conv2d(v, sd[prefix + ".2.weight"], 1, dil, dil, pre=bn(sd, prefix + ".0"), depthwise=True)
        return conv2d(v, sd[prefix + ".3.weight"], 1, 0, 1)

    def r1(v, name):
        v = conv2d(v, sd[name + ".0.weight"], 1, 1, 1)
        for k in range(4):
            v = dws(v, f"{name}.{k + 1}", 2 ** (k + 1))
        return v

    rl = r1(left, "refinement1_left")
    rd = r1(pred3, "refinement1_disp")
    v = np.concatenate([rl, rd], 1)
    v = conv2d(v, sd["refinement2.0.2.weight"], 1, 8, 8, pre=bn(sd, "refinement2.0.0"))
    for i, k in enumerate(reversed(range(4))):
        v = dws(v, f"refinement2.{i + 1}", 2 ** k)
    v = conv2d(v, sd["refinement2.5.weight"], 1, 1, 1)
    return bn_add_relu(v, None, pred3, False)        # pred[2] + disp_up (same-size resize is the identity)


def forward(left, right, sd, maxdisplist=(24, 5, 5), feature_fp16=False):
    """LWSNet.forward (models.py:106-164) entirely through the C restatement."""
    left, right = _c(left), _c(right)
    B, _, H, W = left.shape
    both = feature_extraction(np.concatenate([left, right]), sd)
    fl = [f[:B] for f in both]
    fr = [f[B:] for f in both]
    pred = disparity_stages(fl, fr, H, W, sd, maxdisplist, feature_fp16=feature_fp16)
    pred.append(refine(left, pred[2], sd))
    return pred
