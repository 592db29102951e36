"""Image / disparity file I/O used by the inference CLI (PIL + numpy; the reference uses cv2, which is absent here).

* input pipeline: /root/reference/inference.py:90-103 (crop bottom-right 368x1232, RGB, /255, ImageNet normalise)
* output pipeline: inference.py:113-122 (float -> uint8 cast, JET colour map, PNG)
* PFM reader: /root/reference/dataloader/readpfm.py:6-42
"""
from __future__ import annotations

import re

import numpy as np
from PIL import Image

from .synth import IMAGENET_MEAN, IMAGENET_STD

CROP_H, CROP_W = 368, 1232          # inference.py:94


def load_rgb(path):
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)


def crop_bottom_right(img, th=CROP_H, tw=CROP_W):
    """inference.py:96-100: images smaller than the crop are skipped (returns None)."""
    h, w = img.shape[:2]
    if h < th or w < tw:
        return None
    return img[h - th:h, w - tw:w]


def to_input(rgb_u8):
    """HWC uint8 RGB -> [3,H,W] float32, ToTensor + Normalize(imagenet) (inference.py:83-85,102)."""
    x = rgb_u8.astype(np.float32) / np.float32(255.0)
    x = (x - IMAGENET_MEAN) / IMAGENET_STD
    return np.ascontiguousarray(x.transpose(2, 0, 1), dtype=np.float32)


def jet_lut():
    """256-entry JET colour table (RGB).  MATLAB-jet formula; cv2.COLORMAP_JET interpolates a 64-node table of the
    same map, so individual entries may differ by a couple of levels (visual output only)."""
    v = np.arange(256, dtype=np.float64) / 255.0
    r = np.clip(1.5 - np.abs(4.0 * v - 3.0), 0, 1)
    g = np.clip(1.5 - np.abs(4.0 * v - 2.0), 0, 1)
    b = np.clip(1.5 - np.abs(4.0 * v - 1.0), 0, 1)
    return np.stack([r, g, b], 1).__mul__(255.0).round().astype(np.uint8)


def disparity_to_color(disp):
    """inference.py:114-115: `.astype(np.uint8)` (C cast: truncation, wrap-around outside 0..255) then JET."""
    d8 = np.asarray(disp, dtype=np.float32).astype(np.int64).astype(np.uint8)
    return jet_lut()[d8]


def save_png(path, rgb):
    Image.fromarray(rgb).save(path)


def read_pfm(path):
    """Returns (data, scale); data is [H,W] or [H,W,3] float32, top row first."""
    with open(path, "rb") as f:
        header = f.readline().rstrip()
        if header == b"PF":
            color = True
        elif header == b"Pf":
            color = False
        else:
            raise ValueError("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise ValueError("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)
