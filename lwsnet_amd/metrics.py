"""Evaluation metrics of the reference."""
import numpy as np


def error_3px(disp, gt, maxdisp=192):
    """KITTI 3-pixel error, /root/reference/finetune.py:212-219: over 0 < gt < maxdisp, the fraction of pixels with
    |d - gt| > 3 and |d - gt| / gt > 0.05."""
    disp = np.asarray(disp, dtype=np.float64)
    gt = np.asarray(gt, dtype=np.float64)
    mask = (gt > 0) & (gt < maxdisp)
    err = np.abs(disp - gt)
    bad = (err[mask] > 3.0) & (err[mask] / gt[mask] > 0.05)
    return float(bad.sum()) / float(mask.sum())


def end_point_error(disp, gt, maxdisp=192):
    """SceneFlow EPE, /root/reference/train.py:179 (mask = gt < maxdisp) and :190 (mean |d - gt| over the mask)."""
    disp = np.asarray(disp, dtype=np.float64)
    gt = np.asarray(gt, dtype=np.float64)
    mask = gt < maxdisp
    return float(np.abs(disp[mask] - gt[mask]).mean())
