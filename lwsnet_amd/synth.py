"""Seeded synthetic stereo pairs (numpy only; no datasets travel to the GPU box).

``make_pair`` builds a band-limited left image, a smooth ground-truth disparity
field inside 0..192 px, and the right image as the left one resampled along x,
then normalises both like the reference input pipeline
(/root/reference/inference.py:83-85,102-103; dataloader/dataloader.py:10-11).
``make_noise_pair`` is the adversarial-numerics case (white noise).
"""
from __future__ import annotations

import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def check_size(H, W, maxdisp0=24):
    """Sizes the network accepts: the 2D hourglass skip-adds
    (/root/reference/models/submodules.py:103,182) need ceil(H/2), ceil(W/2)
    divisible by 4 (so H = 8k or 8k-1: the stem convolution, submodules.py:118-125,
    gives ceil(H/2)), and the stage-1 volume (models/models.py:72) needs the 1/8 map
    to be at least maxdisplist[0] wide."""
    h2, w2 = (H + 1) // 2, (W + 1) // 2
    if H <= 0 or W <= 0 or h2 % 4 or w2 % 4:
        raise ValueError(f"unsupported input size {H}x{W}: ceil(H/2) and ceil(W/2) must be divisible by 4")
    if w2 // 4 < maxdisp0:
        raise ValueError(f"unsupported input size {H}x{W}: the 1/8 map is {w2 // 4} wide, must be >= maxdisplist[0]={maxdisp0}")


def _box_blur(a, k=9):
    """Separable k x k box blur with edge replication, float64, along the last two axes."""
    r = k // 2
    for axis in (-1, -2):
        pad = [(0, 0)] * a.ndim
        pad[axis] = (r, r)
        p = np.pad(a, pad, mode="edge")
        c = np.cumsum(p, axis=axis)
        z = np.zeros_like(np.take(c, [0], axis=axis))
        c = np.concatenate([z, c], axis=axis)
        n = a.shape[axis]
        hi = np.take(c, np.arange(k, k + n), axis=axis)
        lo = np.take(c, np.arange(0, n), axis=axis)
        a = (hi - lo) / k
    return a


def gt_disparity(H, W):
    y = np.arange(H, dtype=np.float64)[:, None]
    x = np.arange(W, dtype=np.float64)[None, :]
    return 10.0 + 60.0 * y / H + 6.0 * np.sin(2.0 * np.pi * x / W * 3.0)


def _normalise(img01):
    return ((img01.astype(np.float32) - IMAGENET_MEAN[:, None, None]) / IMAGENET_STD[:, None, None]).astype(np.float32)


def make_pair(H, W, index=0):
    """Returns (left, right, gt) float32: [3,H,W], [3,H,W], [H,W]."""
    rng = np.random.default_rng(1234 + int(index))
    f = rng.random((3, H, W))
    f = _box_blur(_box_blur(f))
    lo, hi = f.min(), f.max()
    left = (f - lo) / (hi - lo)
    g = gt_disparity(H, W)
    xs = np.arange(W, dtype=np.float64)[None, :] - g          # sample left at x - g
    x0 = np.floor(xs)
    lam = xs - x0
    x0 = x0.astype(np.int64)
    x1 = x0 + 1
    rows = np.arange(H)[:, None]

    def tap(xi):
        ok = (xi >= 0) & (xi < W)
        v = left[:, rows, np.clip(xi, 0, W - 1)]
        return np.where(ok[None], v, 0.0)

    right = tap(x0) * (1.0 - lam)[None] + tap(x1) * lam[None]
    return (np.ascontiguousarray(_normalise(left)), np.ascontiguousarray(_normalise(right)),
            np.ascontiguousarray(g, dtype=np.float32))


def make_noise_pair(H, W, index=0):
    rng = np.random.default_rng(4321 + int(index))
    left = rng.standard_normal((3, H, W)).astype(np.float32)
    right = rng.standard_normal((3, H, W)).astype(np.float32)
    return left, right


def make_batch(B, H, W, first_index=0):
    """[B,3,H,W] left/right float32 batches of seeded synthetic pairs."""
    ls, rs = [], []
    for i in range(B):
        l, r, _ = make_pair(H, W, first_index + i)
        ls.append(l)
        rs.append(r)
    return np.ascontiguousarray(np.stack(ls)), np.ascontiguousarray(np.stack(rs))
