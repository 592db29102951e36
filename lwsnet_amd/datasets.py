"""Dataset plumbing of the reference, host side only (numpy + PIL, no framework).

* `kitti2015_lists`   <- /root/reference/dataloader/kitti2015load.py:6-34
* `sceneflow_lists`   <- /root/reference/dataloader/sceneflow.py:37-122
* `StereoPairs`       <- /root/reference/dataloader/dataloader.py:27-95 (train crop 256x512; evaluation crops
                          368x1232 for KITTI and 544x960 for SceneFlow; /255, CHW, ImageNet normalise)

The behaviour is the reference's, including its quirks, because list ORDER and crop GEOMETRY decide which pixels the
published metrics were computed on:
  - KITTI keeps only the `*_10.png` frames; the 40 validation frames come from a split file (sorted) or from a shuffle
    of 0..199; training frames keep directory-listing order;
  - the SceneFlow "driving" subset lists the 15 mm focal-length folder twice and the 35 mm one never (sceneflow.py:106);
  - SceneFlow evaluation crops a 540-row frame to 544 rows from the bottom, i.e. 4 all-zero rows are added on top
    (PIL pads a crop box that leaves the image) while the ground truth keeps 540 rows -- train.py:189 drops the 4 rows.
Everything returns plain numpy arrays ([3,H,W] float32 images, [H,W] float32 disparity) that `LWSNet` accepts as is.
"""
from __future__ import annotations

import os
import random

import numpy as np
from PIL import Image

from .imageio import read_pfm
from .synth import IMAGENET_MEAN, IMAGENET_STD

IMAGE_SUFFIXES = (".jpg", ".JPG", ".jpeg", ".JPEG", ".png", ".PNG", ".ppm", ".PPM", ".bmp", ".BMP")
TRAIN_CROP = (256, 512)            # dataloader.py:62
KITTI_EVAL_CROP = (368, 1232)      # dataloader.py:81-83
SCENEFLOW_EVAL_CROP = (544, 960)   # dataloader.py:85-86


def _is_image(name):
    return name.endswith(IMAGE_SUFFIXES)


def kitti2015_lists(root, split_file=None, shuffle=np.random.shuffle):
    """Returns (left_train, right_train, disp_train, left_val, right_val, disp_val) path lists."""
    frames = [n for n in os.listdir(os.path.join(root, "image_2")) if "_10" in n]
    if split_file is None:
        order = np.arange(200)
        shuffle(order)
        val_ids = list(order[:40])
    else:
        with open(split_file) as f:
            val_ids = sorted(int(line.strip()) for line in f if len(line) > 0 and line.strip())
    val = [f"{int(i):06d}_10.png" for i in val_ids]
    held_out = set(val)
    train = [n for n in frames if n not in held_out]

    def paths(folder, names):
        return [os.path.join(root, folder, n) for n in names]

    return (paths("image_2/", train), paths("image_3/", train), paths("disp_occ_0/", train),
            paths("image_2/", val), paths("image_3/", val), paths("disp_occ_0/", val))


def sceneflow_lists(root):
    """Returns (left_train, right_train, disp_train, left_test, right_test, disp_test) for the SceneFlow layout
    (monkaa + flyingthings3d TRAIN/TEST + driving), clean pass."""
    root = root.rstrip("/") + "/"
    top = [d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))]
    img_sets = [d for d in top if "frames_cleanpass" in d]
    disp_sets = [d for d in top if "disparity" in d]
    tr_l, tr_r, tr_d, te_l, te_r, te_d = [], [], [], [], [], []

    # monkaa: every scene folder; left images (+ their .pfm), then right images
    mk_img = root + next(d for d in img_sets if "monkaa" in d)
    mk_disp = root + next(d for d in disp_sets if "monkaa" in d)
    for scene in os.listdir(mk_img):
        for name in os.listdir(f"{mk_img}/{scene}/left/"):
            if _is_image(name):
                tr_l.append(f"{mk_img}/{scene}/left/{name}")
                tr_d.append(f"{mk_disp}/{scene}/left/{name.split('.')[0]}.pfm")
        for name in os.listdir(f"{mk_img}/{scene}/right/"):
            if _is_image(name):
                tr_r.append(f"{mk_img}/{scene}/right/{name}")

    # flyingthings3d: TRAIN -> training lists, TEST -> test lists; the left listing drives all three lists and the
    # disparity path is appended for every entry of it
    fl_img = root + next(d for d in img_sets if d == "frames_cleanpass")
    fl_disp = root + next(d for d in disp_sets if d == "frames_disparity")
    for split, (L, R, D) in (("TRAIN", (tr_l, tr_r, tr_d)), ("TEST", (te_l, te_r, te_d))):
        for part in ("A", "B", "C"):
            for seq in os.listdir(f"{fl_img}/{split}/{part}"):
                base = f"{fl_img}/{split}/{part}/{seq}"
                for name in os.listdir(base + "/left/"):
                    if _is_image(base + "/left/" + name):
                        L.append(base + "/left/" + name)
                    D.append(f"{fl_disp}/{split}/{part}/{seq}/left/{name.split('.')[0]}.pfm")
                    if _is_image(base + "/right/" + name):
                        R.append(base + "/right/" + name)

    # driving: the reference iterates ['15mm_focallength', '15mm_focallength'] (sceneflow.py:106): 15 mm twice
    dr_img = root + next(d for d in img_sets if "driving" in d) + "/"
    dr_disp = root + next(d for d in disp_sets if "driving" in d)
    for focal in ("15mm_focallength", "15mm_focallength"):
        for direction in ("scene_backwards", "scene_forwards"):
            for speed in ("fast", "slow"):
                base = f"{dr_img}{focal}/{direction}/{speed}"
                for name in os.listdir(base + "/left/"):
                    if _is_image(base + "/left/" + name):
                        tr_l.append(base + "/left/" + name)
                    tr_d.append(f"{dr_disp}/{focal}/{direction}/{speed}/left/{name.split('.')[0]}.pfm")
                    if _is_image(base + "/right/" + name):
                        tr_r.append(base + "/right/" + name)
    return tr_l, tr_r, tr_d, te_l, te_r, te_d


def _normalise_chw(rgb01):
    """Transpose() + Normalize(imagenet) of dataloader.py:42-43 on an HWC float image in 0..1."""
    x = (rgb01 - IMAGENET_MEAN) / IMAGENET_STD
    return np.ascontiguousarray(x.transpose(2, 0, 1), dtype=np.float32)


class StereoPairs:
    """Indexable dataset: `ds[i] -> (left [3,h,w], right [3,h,w], disparity [h',w'])`, float32 numpy."""

    def __init__(self, left, right, left_disparity, training=True, kitti_set=True, rng=random):
        assert len(left) == len(right) == len(left_disparity)
        self.left, self.right, self.disp = list(left), list(right), list(left_disparity)
        self.training, self.kitti_set, self.rng = training, kitti_set, rng

    def __len__(self):
        return len(self.left)

    def _disparity(self, path):
        if self.kitti_set:                                     # 16-bit PNG, value / 256 (dataloader.py:54-55)
            return np.ascontiguousarray(Image.open(path), dtype=np.float32) / 256
        data, _scale = read_pfm(path)                          # dataloader.py:57-58
        return np.ascontiguousarray(data, dtype=np.float32)

    def __getitem__(self, index):
        li = Image.open(self.left[index]).convert("RGB")
        ri = Image.open(self.right[index]).convert("RGB")
        d = self._disparity(self.disp[index])
        w, h = li.size
        if self.training:                                      # random 256x512 window (dataloader.py:60-75)
            th, tw = TRAIN_CROP
            x1 = self.rng.randint(0, w - tw)
            y1 = self.rng.randint(0, h - th)
            box = (x1, y1, x1 + tw, y1 + th)
            d = d[y1:y1 + th, x1:x1 + tw]
        elif self.kitti_set:                                   # bottom-right 368x1232, ground truth cropped alike
            th, tw = KITTI_EVAL_CROP
            box = (w - tw, h - th, w, h)
            d = d[h - th:h, w - tw:w]
        else:                                                  # bottom-right 544x960: PIL pads the rows above the image
            th, tw = SCENEFLOW_EVAL_CROP                       # with zeros; the ground truth keeps its 540 rows
            box = (w - tw, h - th, w, h)
        lf = np.array(li.crop(box), dtype=np.float32) / 255
        rf = np.array(ri.crop(box), dtype=np.float32) / 255
        return _normalise_chw(lf), _normalise_chw(rf), d
