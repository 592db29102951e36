"""TEST INFRASTRUCTURE ONLY -- literal CPU restatement of the reference forward pass.

PARITY UNPINNED: the reference (/root/reference/models/models.py, submodules.py)
is Python on PaddlePaddle 2.0.0rc0 (paddle_env.yml:149); paddle is not
installable here and the reference holds no tests, golden vectors or numeric
fixtures for this path.  This file restates
``LWSNet.forward`` op by op with torch-CPU functionals whose documented
semantics coincide with the Paddle defaults the reference relies on
(SURVEY.md appendix B).  Nothing in the shipped package may import it: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, as the checker.

``dtype=torch.float32`` mirrors the reference arithmetic; ``torch.float64`` is
the "truth" used to tell fp32 noise from bugs.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5

# ---- the op defaults this restatement BETS on (from memory of Paddle 2.0, not verifiable offline), kept
# as switches so that the size of each bet can be measured (tools/oracle_sensitivity.py, profiles/r04/oracle_sensitivity.txt).
# The first value of each line is the default = what every parity test and the C oracle / HIP kernels implement.
#   align_mode   0: F.interpolate(bilinear, align_corners=False) samples at half-pixel centres, src = ratio*(dst+0.5)-0.5
#                1: Paddle 1.x / 2.0-beta default, src = ratio*dst
#   grid_unnorm  "cuda": grid_sample un-normalises as ((g+1)/2)*(size-1)   (Paddle's CUDA kernel, torch)
#                "cpu":  (g+1)*((size-1)*0.5)                                 (Paddle's CPU kernel)
#   scalar_div   "reciprocal": tensor / python_scalar is a scale op by fp32(1/c)
#                "divide":     a true elementwise division
VARIANT = {"align_mode": 0, "grid_unnorm": "cuda", "scalar_div": "reciprocal"}


class variant:
    """with variant(align_mode=1): ...  -- run the restatement under another reading of Paddle's defaults."""

    def __init__(self, **kw):
        unknown = set(kw) - set(VARIANT)
        if unknown:
            raise KeyError(f"unknown oracle variant switch(es): {sorted(unknown)}")
        self.kw = kw

    def __enter__(self):
        self.saved = dict(VARIANT)
        VARIANT.update(self.kw)
        return self

    def __exit__(self, *a):
        VARIANT.clear()
        VARIANT.update(self.saved)


def interp_bilinear(x, size, align_mode=0):
    """Paddle's bilinear F.interpolate with align_corners=False.  align_mode 0 = half-pixel centres (torch's
    align_corners=False); align_mode 1 = src = ratio * dst (clamped taps), written out separably."""
    if align_mode == 0:
        return F.interpolate(x, size=list(size), mode="bilinear", align_corners=False)

    def axis(n_in, n_out):
        ratio = torch.tensor(float(n_in), dtype=x.dtype) / torch.tensor(float(n_out), dtype=x.dtype)
        src = ratio * torch.arange(n_out, dtype=x.dtype)
        i0 = src.floor().clamp(max=n_in - 1)
        lam = src - i0
        i0 = i0.long()
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, lam

    y0, y1, ly = axis(x.shape[2], size[0])
    x0, x1, lx = axis(x.shape[3], size[1])
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly)[:, None] + bot * ly[:, None]


def grid_sample_bilinear(x, grid, unnorm="cuda"):
    """F.grid_sample(bilinear, zeros, align_corners=True).  "cuda" is torch's own kernel; "cpu" re-derives the pixel
    coordinates as (g+1)*((size-1)*0.5) and gathers the four taps (nw, ne, sw, se summed in that order)."""
    if unnorm == "cuda":
        return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    B, C, H, W = x.shape
    ix = (grid[..., 0] + 1.0) * ((W - 1) * 0.5)
    iy = (grid[..., 1] + 1.0) * ((H - 1) * 0.5)
    x0, y0 = ix.floor(), iy.floor()
    x1, y1 = x0 + 1, y0 + 1
    flat = x.reshape(B, C, H * W)
    out = torch.zeros((B, C) + tuple(ix.shape[1:]), dtype=x.dtype)
    for xs, ys, wgt in ((x0, y0, (x1 - ix) * (y1 - iy)), (x1, y0, (ix - x0) * (y1 - iy)),
                        (x0, y1, (x1 - ix) * (iy - y0)), (x1, y1, (ix - x0) * (iy - y0))):
        ok = (xs >= 0) & (xs <= W - 1) & (ys >= 0) & (ys <= H - 1)
        idx = (ys.clamp(0, H - 1) * W + xs.clamp(0, W - 1)).long().reshape(B, 1, -1).expand(B, C, -1)
        tap = flat.gather(2, idx).reshape(out.shape)
        out = out + tap * (wgt * ok.to(x.dtype)).unsqueeze(1)
    return out


def _t(sd, key, dtype):
    return torch.as_tensor(np.asarray(sd[key]), dtype=dtype)


def _bn(x, sd, prefix, dtype):
    # nn.BatchNorm2D/3D in eval mode, epsilon=1e-5 (submodules.py:18,196,226)
    return F.batch_norm(x, _t(sd, prefix + "._mean", dtype), _t(sd, prefix + "._variance", dtype),
                        _t(sd, prefix + ".weight", dtype), _t(sd, prefix + ".bias", dtype),
                        training=False, eps=EPS)


def _scale(x, c, dtype):
    """Paddle lowers ``tensor / python_scalar`` to a scale op by the reciprocal
    (SURVEY.md appendix B); the reciprocal is rounded to the tensor dtype."""
    if VARIANT["scalar_div"] == "divide":
        return x / torch.tensor(float(c), dtype=dtype)
    r = torch.tensor(1.0, dtype=dtype) / torch.tensor(float(c), dtype=dtype)
    return x * r


# ----------------------------------------------------------------------------
# 2D feature extractor (submodules.py:5-33, 35-109, 113-188)
# ----------------------------------------------------------------------------
def _convbn(x, sd, prefix, stride, pad, dil, dtype):
    # convbn(): padding = dilation if dilation > 1 else padding  (submodules.py:14)
    p = dil if dil > 1 else pad
    x = F.conv2d(x, _t(sd, prefix + ".0.weight", dtype), None, stride, p, dil)
    return _bn(x, sd, prefix + ".1", dtype)


def _deconvbn(x, sd, prefix, dtype):
    # Conv2DTranspose(k=3, s=2, p=1, output_padding=1)  (submodules.py:25-32)
    x = F.conv_transpose2d(x, _t(sd, prefix + ".0.weight", dtype), None, stride=2, padding=1, output_padding=1)
    return _bn(x, sd, prefix + ".1", dtype)


def feature_extraction(x, sd, dtype=torch.float32):
    fe = "feature_extraction"
    o = F.relu(_convbn(x, sd, f"{fe}.dres0.0", 2, 1, 2, dtype))          # :118-126
    o = F.relu(_convbn(o, sd, f"{fe}.dres0.2", 1, 1, 4, dtype))          # :127-135
    r = F.relu(_convbn(o, sd, f"{fe}.dres1.0", 1, 1, 2, dtype))          # :137-145
    r = _convbn(r, sd, f"{fe}.dres1.2", 1, 1, 2, dtype)                  # :146-153 (no ReLU)
    o = r + o                                                            # :179
    hg = f"{fe}.dres2"
    c1 = F.relu(_convbn(o, sd, f"{hg}.conv1.0", 2, 1, 1, dtype))         # :96
    pre = F.relu(_convbn(c1, sd, f"{hg}.conv2.0", 1, 1, 1, dtype))       # :97
    c3 = F.relu(_convbn(pre, sd, f"{hg}.conv3.0", 2, 1, 1, dtype))       # :99
    f8 = F.relu(_convbn(c3, sd, f"{hg}.conv4.0", 1, 1, 1, dtype))        # :100-101
    f4 = F.relu(_deconvbn(f8, sd, f"{hg}.conv5", dtype) + pre)           # :103-104
    c6 = _deconvbn(f4, sd, f"{hg}.conv6", dtype)                         # :106
    o = c6 + o                                                           # :182
    o = F.relu(_convbn(o, sd, f"{fe}.classif1.0", 1, 1, 1, dtype))       # :157-165
    f2 = F.conv2d(o, _t(sd, f"{fe}.classif1.2.weight", dtype), None, 1, 1)  # :166-172
    return [f8, f4, f2]


# ----------------------------------------------------------------------------
# hot path (models.py:28-104, 167-179; submodules.py:190-221)
# ----------------------------------------------------------------------------
def warp(x, disp, dtype=torch.float32):
    """models.py:28-55.  grid_sample defaults: bilinear, zeros, align_corners=True."""
    B, C, H, W = x.shape
    xx = torch.arange(0, W, dtype=dtype).reshape(1, -1).expand(H, W)
    yy = torch.arange(0, H, dtype=dtype).reshape(-1, 1).expand(H, W)
    xx = xx.reshape(1, 1, H, W).expand(B, 1, H, W)
    yy = yy.reshape(1, 1, H, W).expand(B, 1, H, W)
    vgrid = torch.cat((xx, yy), 1).clone()
    vgrid[:, :1] = vgrid[:, :1] - disp
    vgrid[:, 0] = _scale(2.0 * vgrid[:, 0], max(W - 1, 1), dtype) - 1.0
    vgrid[:, 1] = _scale(2.0 * vgrid[:, 1], max(H - 1, 1), dtype) - 1.0
    vgrid = vgrid.permute(0, 2, 3, 1)
    return grid_sample_bilinear(x, vgrid, VARIANT["grid_unnorm"])


def build_volume_2d(feat_l, feat_r, maxdisp, dtype=torch.float32):
    """models.py:58-76 (stride is always 1 on the path, :134)."""
    B, C, h, w = feat_l.shape
    cost = torch.zeros((B, maxdisp, h, w), dtype=dtype)
    for i in range(maxdisp):
        if i > 0:
            cost[:, i, :, :i] = feat_l[:, :, :, :i].abs().sum(1)
            cost[:, i, :, i:] = (feat_l[:, :, :, i:] - feat_r[:, :, :, :-i]).abs().sum(1)
        else:
            cost[:, i, :, :] = (feat_l - feat_r).abs().sum(1)
    return cost


def build_volume_2d3(feat_l, feat_r, maxdisp, disp, dtype=torch.float32):
    """models.py:78-104.  The 9x expansion is done one sample at a time to bound memory."""
    B, C, h, w = feat_l.shape
    n = 2 * maxdisp - 1
    shift = torch.arange(-maxdisp + 1, maxdisp, dtype=dtype).reshape(n, 1, 1, 1)
    outs = []
    for b in range(B):
        bd = disp[b:b + 1].expand(n, 1, h, w) - shift          # :93
        fl = feat_l[b:b + 1].expand(n, C, h, w)
        fr = feat_r[b:b + 1].expand(n, C, h, w)
        c = (fl - warp(fr, bd, dtype)).abs().sum(1)            # :101
        outs.append(c)
    return torch.stack(outs, 0)


def post_3dconvs(cost5, sd, stage, dtype=torch.float32):
    """submodules.py:216-221: [BN3D-ReLU-Conv3D] x (layers+2)."""
    y = cost5
    j = 0
    while f"volume_postprocess.{stage}.{j}.2.weight" in sd:
        p = f"volume_postprocess.{stage}.{j}"
        y = F.relu(_bn(y, sd, p + ".0", dtype))
        y = F.conv3d(y, _t(sd, p + ".2.weight", dtype), None, 1, 1)
        j += 1
    return y


def disparity_regression(prob, start, end, dtype=torch.float32):
    """models.py:167-179."""
    disp = torch.arange(start, end, dtype=dtype).reshape(1, -1, 1, 1)
    return torch.sum(prob * disp, 1, keepdim=True)


def _interp(x, size):
    # F.interpolate(mode="bilinear"): align_corners=False, align_mode=0 (half-pixel) unless VARIANT says otherwise
    return interp_bilinear(x, size, VARIANT["align_mode"])


def disparity_stages(feats_l, feats_r, H, W, sd, maxdisplist, dtype=torch.float32, return_costs=False):
    """models.py:115-156 for the three volume stages.  Returns [pred1, pred2, pred3]."""
    pred, costs = [], []
    for scale in range(3):
        fl, fr = feats_l[scale], feats_r[scale]
        h, w = fl.shape[2], fl.shape[3]
        if scale > 0:
            wflow = _scale(_interp(pred[scale - 1], [h, w]) * float(h), H, dtype)       # :119-121
            cost = build_volume_2d3(fl, fr, maxdisplist[scale], wflow, dtype)            # :123
        else:
            cost = build_volume_2d(fl, fr, maxdisplist[0], dtype)                         # :131
        raw = cost
        cost = cost.unsqueeze(1)
        cost = post_3dconvs(cost, sd, scale, dtype) + cost                                # :137
        cost = cost.squeeze(1)
        if return_costs:
            costs.append((raw, cost))
        p = F.softmax(-cost, dim=1)
        if scale == 0:
            low = disparity_regression(p, 0, maxdisplist[0], dtype)                       # :142
        else:
            low = disparity_regression(p, -maxdisplist[scale] + 1, maxdisplist[scale], dtype)  # :151
        low = _scale(low * float(H), low.shape[2], dtype)                                  # :145,153
        up = _interp(low, [H, W])                                                          # :146,154
        pred.append(up if scale == 0 else up + pred[scale - 1])                            # :148,156
    return (pred, costs) if return_costs else pred


# ----------------------------------------------------------------------------
# refinement (submodules.py:223-327; models.py:158-162)
# ----------------------------------------------------------------------------
def _dws_block(x, sd, prefix, dil, dtype):
    # preconv2d_depthseperated: BN, ReLU, depthwise 3x3 (dilated), pointwise 1x1
    x = F.relu(_bn(x, sd, prefix + ".0", dtype))
    c = x.shape[1]
    x = F.conv2d(x, _t(sd, prefix + ".2.weight", dtype), None, 1, dil if dil > 1 else 1, dil, groups=c)
    return F.conv2d(x, _t(sd, prefix + ".3.weight", dtype), None, 1, 0)


def refinement1(x, sd, name, dtype=torch.float32):
    x = F.conv2d(x, _t(sd, name + ".0.weight", dtype), None, 1, 1)
    for k in range(4):
        x = _dws_block(x, sd, f"{name}.{k + 1}", 2 ** (k + 1), dtype)
    return x


def refinement2(x, sd, dtype=torch.float32):
    x = F.relu(_bn(x, sd, "refinement2.0.0", dtype))
    x = F.conv2d(x, _t(sd, "refinement2.0.2.weight", dtype), None, 1, 8, 8)
    for i, k in enumerate(reversed(range(4))):
        x = _dws_block(x, sd, f"refinement2.{i + 1}", 2 ** k, dtype)
    return F.conv2d(x, _t(sd, "refinement2.5.weight", dtype), None, 1, 1)


def refine(left, pred3, sd, dtype=torch.float32):
    """models.py:158-162."""
    H, W = left.shape[2], left.shape[3]
    rl = refinement1(left, sd, "refinement1_left", dtype)
    rd = refinement1(pred3, sd, "refinement1_disp", dtype)
    d = refinement2(torch.cat([rl, rd], 1), sd, dtype)
    return pred3 + _interp(d, [H, W])


def forward(left, right, sd, maxdisplist=(24, 5, 5), dtype=torch.float32):
    """LWSNet.forward (models.py:106-164): list of 4 tensors [B,1,H,W]."""
    left = torch.as_tensor(np.asarray(left), dtype=dtype)
    right = torch.as_tensor(np.asarray(right), dtype=dtype)
    H, W = left.shape[2], left.shape[3]
    with torch.no_grad():
        fl = feature_extraction(left, sd, dtype)
        fr = feature_extraction(right, sd, dtype)
        pred = disparity_stages(fl, fr, H, W, sd, list(maxdisplist), dtype)
        pred.append(refine(left, pred[2], sd, dtype))
    return pred


def error_3px(disp, gt, maxdisp=192):
    """/root/reference/finetune.py:212-219."""
    disp = np.asarray(disp, dtype=np.float64)
    gt = np.asarray(gt, dtype=np.float64)
    mask = (gt > 0) & (gt < maxdisp)
    err = np.abs(disp - gt)
    bad = (err[mask] > 3.0) & (err[mask] / gt[mask] > 0.05)
    return float(bad.sum()) / float(mask.sum())
