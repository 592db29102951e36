"""Per-op wrappers over the C ABI for torch ROCm tensors (device memory + stream plumbing only).

Each function mirrors one reference symbol (see include/lwsnet_hip.h for file:line) and
launches the hand-written HIP kernel on torch's current stream.  No CPU path exists:
tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a torch tensor on a HIP device (the disparity path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def volume_l1_shift(feat_l, feat_r, maxdisp):
    """LWSNet._build_volume_2d (models/models.py:58-76)."""
    L, R = _dev(feat_l, "feat_l"), _dev(feat_r, "feat_r")
    if L.shape != R.shape or L.dim() != 4:
        raise ValueError(f"feat_l/feat_r must both be [B,C,h,w]; got {tuple(L.shape)} and {tuple(R.shape)}")
    B, C, h, w = L.shape
    cost = torch.empty((B, maxdisp, h, w), device=L.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(L.device):
        _lib.check(lib.lws_volume_l1_shift(_ptr(L), _ptr(R), _ptr(cost), B, C, h, w, int(maxdisp), _stream()),
                   "lws_volume_l1_shift")
    return cost


def volume_l1_warp(feat_l, feat_r, prev_disp, maxdisp, return_wflow=False):
    """forward() glue (models/models.py:119-121) + _build_volume_2d3 (:78-104) + warp (:28-55)."""
    L, R, P = _dev(feat_l, "feat_l"), _dev(feat_r, "feat_r"), _dev(prev_disp, "prev_disp")
    if L.shape != R.shape or L.dim() != 4:
        raise ValueError(f"feat_l/feat_r must both be [B,C,h,w]; got {tuple(L.shape)} and {tuple(R.shape)}")
    B, C, h, w = L.shape
    if P.dim() != 4 or P.shape[0] != B or P.shape[1] != 1:
        raise ValueError(f"prev_disp must be [B,1,H,W]; got {tuple(P.shape)}")
    H, W = P.shape[2], P.shape[3]
    cost = torch.empty((B, 2 * maxdisp - 1, h, w), device=L.device, dtype=torch.float32)
    wflow = torch.empty((B, h, w), device=L.device, dtype=torch.float32) if return_wflow else None
    lib = _lib.load()
    with torch.cuda.device(L.device):
        _lib.check(lib.lws_volume_l1_warp(_ptr(L), _ptr(R), _ptr(P), _ptr(cost), _ptr(wflow), B, C, h, w, H, W,
                                          int(maxdisp), _stream()), "lws_volume_l1_warp")
    return (cost, wflow) if return_wflow else cost


def conv3d_stack(handle, stage, cost):
    """volume_postprocess[stage](cost) + cost (models/models.py:136-138)."""
    c = _dev(cost, "cost")
    if c.dim() != 4:
        raise ValueError(f"cost must be [B,D,h,w]; got {tuple(c.shape)}")
    B, D, h, w = c.shape
    out = torch.empty_like(c)
    lib = _lib.load()
    with torch.cuda.device(c.device):
        _lib.check(lib.lws_conv3d_stack(handle, int(stage), _ptr(c), _ptr(out), B, D, h, w, _stream()),
                   "lws_conv3d_stack")
    return out


def softargmin(cost, start):
    """F.softmax(-cost, 1) + disparity_regression (models/models.py:142,151-152,167-179)."""
    c = _dev(cost, "cost")
    B, D, h, w = c.shape
    low = torch.empty((B, h, w), device=c.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(c.device):
        _lib.check(lib.lws_softargmin(_ptr(c), _ptr(low), B, D, h, w, float(start), _stream()), "lws_softargmin")
    return low


def upsample_add(disp_low, prev, H, W):
    """models/models.py:145-148,153-156."""
    low = _dev(disp_low, "disp_low")
    B, h, w = low.shape
    prev = _dev(prev, "prev") if prev is not None else None
    out = torch.empty((B, 1, H, W), device=low.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(low.device):
        _lib.check(lib.lws_upsample_add(_ptr(low), _ptr(prev), _ptr(out), B, h, w, int(H), int(W), _stream()),
                   "lws_upsample_add")
    return out


def disparity_stages(handle, feats_l, feats_r, H, W):
    """The body of `for scale in range(3)` (models/models.py:115-156): returns [pred1, pred2, pred3]."""
    fl = [_dev(t, f"feats_l[{i}]") for i, t in enumerate(feats_l)]
    fr = [_dev(t, f"feats_r[{i}]") for i, t in enumerate(feats_r)]
    if len(fl) != 3 or len(fr) != 3:
        raise ValueError("feats_l / feats_r must hold the three feature maps (1/8, 1/4, 1/2)")
    B = fl[0].shape[0]
    h2, w2 = (H + 1) // 2, (W + 1) // 2                 # the stem convolution gives ceil(H/2) (submodules.py:118-125)
    want = [(B, 16, h2 // 4, w2 // 4), (B, 16, h2 // 2, w2 // 2), (B, 8, h2, w2)]
    for i in range(3):
        if tuple(fl[i].shape) != want[i] or tuple(fr[i].shape) != want[i]:
            raise ValueError(f"stage {i + 1} features must be {want[i]}; got {tuple(fl[i].shape)} / {tuple(fr[i].shape)}")
    preds = [torch.empty((B, 1, H, W), device=fl[0].device, dtype=torch.float32) for _ in range(3)]
    arr = ctypes.c_void_p * 3
    lib = _lib.load()
    with torch.cuda.device(fl[0].device):
        _lib.check(lib.lws_disparity_stages(handle, arr(*[t.data_ptr() for t in fl]), arr(*[t.data_ptr() for t in fr]),
                                            B, int(H), int(W), arr(*[t.data_ptr() for t in preds]), _stream()),
                   "lws_disparity_stages")
    return preds


def feature_extraction(handle, img):
    """feature_extraction (models/submodules.py:113-188): [N,3,H,W] -> [1/8 (16 ch), 1/4 (16 ch), 1/2 (8 ch)]."""
    x = _dev(img, "img")
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"img must be [N,3,H,W]; got {tuple(x.shape)}")
    N, _, H, W = x.shape
    h2, w2 = (H + 1) // 2, (W + 1) // 2
    f8 = torch.empty((N, 16, h2 // 4, w2 // 4), device=x.device, dtype=torch.float32)
    f4 = torch.empty((N, 16, h2 // 2, w2 // 2), device=x.device, dtype=torch.float32)
    f2 = torch.empty((N, 8, h2, w2), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.lws_feature_extraction(handle, _ptr(x), N, H, W, _ptr(f8), _ptr(f4), _ptr(f2), _stream()),
                   "lws_feature_extraction")
    return [f8, f4, f2]


def refine(handle, left, pred3):
    """models/models.py:158-162: pred4 = pred3 + refinement2(cat(refinement1_left(left), refinement1_disp(pred3)))."""
    l, p3 = _dev(left, "left"), _dev(pred3, "pred3")
    B, _, H, W = l.shape
    if tuple(p3.shape) != (B, 1, H, W):
        raise ValueError(f"pred3 must be {(B, 1, H, W)}; got {tuple(p3.shape)}")
    out = torch.empty_like(p3)
    lib = _lib.load()
    with torch.cuda.device(l.device):
        _lib.check(lib.lws_refine(handle, _ptr(l), _ptr(p3), B, H, W, _ptr(out), _stream()), "lws_refine")
    return out


def stage_outputs(out, B, H, W, device):
    """The four [B,1,H,W] destinations of a forward: `out` is None or a list of four entries, each None (allocated here) or
    a contiguous float32 tensor of exactly that shape on `device` -- a raw pointer is handed to the library, so a strided or
    wrong-sized tensor would be an out-of-bounds device write (shared by ops.forward and ForwardPool.submit)."""
    if out is not None and (not isinstance(out, (list, tuple)) or len(out) != 4):
        raise ValueError("out must be a list of four tensors (or None entries), one per stage")
    preds = []
    for s in range(4):
        t = out[s] if out is not None else None
        if t is None:
            t = torch.empty((B, 1, H, W), device=device, dtype=torch.float32)
        elif not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == torch.device(device) and t.dtype == torch.float32
                  and t.is_contiguous() and tuple(t.shape) == (B, 1, H, W)):
            raise ValueError(f"out[{s}] must be a contiguous float32 tensor of shape {(B, 1, H, W)} on {device}")
        preds.append(t)
    return preds


def forward(handle, left, right, out=None):
    """LWSNet.forward (models/models.py:106-164): list of 4 x [B,1,H,W].  `out` (optional): a list of four destinations;
    entries that are None are allocated here, the others must be contiguous float32 [B,1,H,W] device tensors (a slot of
    a staging buffer, say) and are written in place."""
    l, r = _dev(left, "left"), _dev(right, "right")
    B, _, H, W = l.shape
    preds = stage_outputs(out, B, H, W, l.device)
    arr = ctypes.c_void_p * 4
    lib = _lib.load()
    with torch.cuda.device(l.device):
        _lib.check(lib.lws_forward(handle, _ptr(l), _ptr(r), B, H, W, arr(*[t.data_ptr() for t in preds]), _stream()),
                   "lws_forward")
    return preds


def preprocess_rgb8(rgb_u8, out=None):
    """ToTensor + Normalize(imagenet) of inference.py:83-85,102-103 on the device: rgb_u8 [B,H,W,3] uint8 -> [B,3,H,W] float32,
    bit for bit lwsnet_amd.imageio.to_input (numpy)."""
    from .synth import IMAGENET_MEAN, IMAGENET_STD
    if not isinstance(rgb_u8, torch.Tensor) or not rgb_u8.is_cuda or rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 4 or rgb_u8.shape[3] != 3:
        raise ValueError("rgb_u8 must be a [B,H,W,3] uint8 tensor on a HIP device")
    rgb_u8 = rgb_u8.contiguous()
    B, H, W, _ = rgb_u8.shape
    if out is None:
        out = torch.empty((B, 3, H, W), device=rgb_u8.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, 3, H, W) or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
        raise ValueError("out must be a contiguous [B,3,H,W] float32 device tensor")
    mean = (ctypes.c_float * 3)(*[float(v) for v in IMAGENET_MEAN])
    std = (ctypes.c_float * 3)(*[float(v) for v in IMAGENET_STD])
    with torch.cuda.device(rgb_u8.device):
        _lib.check(_lib.load().lws_preprocess_rgb8(_ptr(rgb_u8), _ptr(out), B, H, W, mean, std, _stream()), "lws_preprocess_rgb8")
    return out


def apply_lut8(disp, lut_dev, out=None):
    """`.astype(np.uint8)` + colour map of inference.py:114-115 on the device: disp (any shape) float32 -> [...,3] uint8 through
    lut_dev ([256,3] uint8 on the device), bit for bit lwsnet_amd.imageio.disparity_to_color."""
    d = _dev(disp, "disp")
    if not isinstance(lut_dev, torch.Tensor) or not lut_dev.is_cuda or lut_dev.dtype != torch.uint8 or lut_dev.numel() != 768:
        raise ValueError("lut_dev must be a [256,3] uint8 tensor on a HIP device")
    if out is None:
        out = torch.empty(tuple(d.shape) + (3,), device=d.device, dtype=torch.uint8)
    elif out.numel() != 3 * d.numel() or out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous():
        raise ValueError("out must be a contiguous uint8 device tensor with 3 bytes per disparity value")
    with torch.cuda.device(d.device):
        _lib.check(_lib.load().lws_apply_lut8(_ptr(d), _ptr(lut_dev.contiguous()), _ptr(out), ctypes.c_int64(d.numel()), _stream()),
                   "lws_apply_lut8")
    return out
