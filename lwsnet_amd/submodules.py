"""2D feature extractor and refinement networks.

These are outside the hot path BASELINE.json names.  Until they get hand-written HIP
kernels they are plumbing: the layer graph of /root/reference/models/submodules.py
(:5-33 convbn/deconvbn, :35-109 hourglass, :113-188 feature_extraction, :223-327
refinement) evaluated with PyTorch-ROCm functional ops on the device tensors.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

EPS = 1e-5


class Params:
    """State dict as device tensors."""

    def __init__(self, sd, device):
        self.t = {k: torch.as_tensor(v, dtype=torch.float32).to(device).contiguous() for k, v in sd.items()}

    def __getitem__(self, key):
        return self.t[key]


def _bn(x, p, prefix):
    return F.batch_norm(x, p[prefix + "._mean"], p[prefix + "._variance"], p[prefix + ".weight"],
                        p[prefix + ".bias"], training=False, eps=EPS)


def _convbn(x, p, prefix, stride, pad, dil):
    x = F.conv2d(x, p[prefix + ".0.weight"], None, stride, dil if dil > 1 else pad, dil)   # submodules.py:14
    return _bn(x, p, prefix + ".1")


def _deconvbn(x, p, prefix):
    x = F.conv_transpose2d(x, p[prefix + ".0.weight"], None, stride=2, padding=1, output_padding=1)
    return _bn(x, p, prefix + ".1")


def feature_extraction(x, p):
    """submodules.py:176-188 -> [1/8 (16 ch), 1/4 (16 ch), 1/2 (8 ch)]."""
    fe = "feature_extraction"
    o = F.relu(_convbn(x, p, f"{fe}.dres0.0", 2, 1, 2))
    o = F.relu(_convbn(o, p, f"{fe}.dres0.2", 1, 1, 4))
    r = F.relu(_convbn(o, p, f"{fe}.dres1.0", 1, 1, 2))
    r = _convbn(r, p, f"{fe}.dres1.2", 1, 1, 2)
    o = r + o
    hg = f"{fe}.dres2"
    c1 = F.relu(_convbn(o, p, f"{hg}.conv1.0", 2, 1, 1))
    pre = F.relu(_convbn(c1, p, f"{hg}.conv2.0", 1, 1, 1))
    c3 = F.relu(_convbn(pre, p, f"{hg}.conv3.0", 2, 1, 1))
    f8 = F.relu(_convbn(c3, p, f"{hg}.conv4.0", 1, 1, 1))
    f4 = F.relu(_deconvbn(f8, p, f"{hg}.conv5") + pre)
    c6 = _deconvbn(f4, p, f"{hg}.conv6")
    o = c6 + o
    o = F.relu(_convbn(o, p, f"{fe}.classif1.0", 1, 1, 1))
    f2 = F.conv2d(o, p[f"{fe}.classif1.2.weight"], None, 1, 1)
    return [f8, f4, f2]


def _dws_block(x, p, prefix, dil):
    x = F.relu(_bn(x, p, prefix + ".0"))
    x = F.conv2d(x, p[prefix + ".2.weight"], None, 1, dil if dil > 1 else 1, dil, groups=x.shape[1])
    return F.conv2d(x, p[prefix + ".3.weight"], None, 1, 0)


def refinement1(x, p, name):
    x = F.conv2d(x, p[name + ".0.weight"], None, 1, 1)
    for k in range(4):
        x = _dws_block(x, p, f"{name}.{k + 1}", 2 ** (k + 1))
    return x


def refinement2(x, p):
    x = F.relu(_bn(x, p, "refinement2.0.0"))
    x = F.conv2d(x, p["refinement2.0.2.weight"], None, 1, 8, 8)
    for i, k in enumerate(reversed(range(4))):
        x = _dws_block(x, p, f"refinement2.{i + 1}", 2 ** k)
    return F.conv2d(x, p["refinement2.5.weight"], None, 1, 1)


def refine(left, pred3, p):
    """models/models.py:158-162 (the same-size bilinear resize at :161 is the identity)."""
    rl = refinement1(left, p, "refinement1_left")
    rd = refinement1(pred3, p, "refinement1_disp")
    d = refinement2(torch.cat([rl, rd], 1), p)
    return pred3 + d
