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
    """OpenCV's 256-entry `COLORMAP_JET` table as RGB rows (inference.py:115 applies it; cv2.imwrite stores BGR as RGB).

    OpenCV linearly interpolates a 64-node table; on the 0..255 grid that is a piecewise-linear integer ramp with
    slope 4 per level and plateaux at 255: blue 128 -> 255 (levels 0..32), green up (32..96), red up / blue down
    (96..160, offset by 2: (2,255,254), (6,255,250), ...), green down (160..223), red down to 128 (224..255).
    Levels 0..127 are pinned byte for byte by the 128 distinct colours of the reference's own
    `reference/{1..4}.png` (tests/golden/jet_reference_colours.npz, tests/test_host_cpu.py); levels 128..255 follow
    from the table's red/blue mirror symmetry `lut[255 - i] = lut[i][::-1]`."""
    i = np.arange(256, dtype=np.int64)

    def ramp(up, down):                       # rising edge 4i+up, falling edge down-4i, clipped to a byte
        return np.minimum(np.clip(4 * i + up, 0, 255), np.clip(down - 4 * i, 0, 255))

    return np.stack([ramp(-382, 1148), ramp(-128, 892), ramp(128, 638)], 1).astype(np.uint8)


_JET = None


def disparity_to_color(disp):
    """inference.py:114-115: `.astype(np.uint8)` (C cast: truncation, wrap-around outside 0..255) then JET."""
    global _JET
    if _JET is None:
        _JET = jet_lut()
    d8 = np.asarray(disp, dtype=np.float32).astype(np.int64).astype(np.uint8)
    return _JET[d8]


def encode_png(rgb):
    """[H,W,3] uint8 -> PNG bytes: 8-bit RGB, filter type 0 on every row, one IDAT of zlib level 1 -- the speed end of what
    `cv2.imwrite(path, img)` (inference.py:120,136; OpenCV's PNG default is compression level 1) stands for.  Lossless like any
    PNG: what is compared is the decoded pixels (tests/test_gpu_parity.py::test_config1_*, tests/test_host_cpu.py).  PIL's own
    encoder at the same level spends twice the time choosing row filters (23 vs 12 ms for a 368x1232 map of noise-like disparities),
    which is most of what a host worker of the pipelined CLI does per pair."""
    import struct
    import zlib
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    if rgb.ndim != 3 or rgb.shape[2] != 3:
        raise ValueError(f"encode_png wants [H,W,3] uint8, got {rgb.shape}")
    h, w, _ = rgb.shape
    raw = np.empty((h, 1 + 3 * w), np.uint8)
    raw[:, 0] = 0                                   # filter type None
    raw[:, 1:] = rgb.reshape(h, 3 * w)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw.tobytes(), 1)) + chunk(b"IEND", b""))


def save_png(path, rgb):
    with open(path, "wb") as f:
        f.write(encode_png(rgb))


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
