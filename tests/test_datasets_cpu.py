"""Dataset plumbing (lwsnet_amd/datasets.py, logger.py) against the behaviour of the reference's dataloader/ and utils/
(file:line in the module docstrings), on small synthetic directory trees."""
import os
import random

import numpy as np
import pytest
from PIL import Image

from lwsnet_amd import datasets as D
from lwsnet_amd.synth import IMAGENET_MEAN, IMAGENET_STD


def _png(path, arr):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def _pfm(path, data):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    h, w = data.shape
    with open(path, "wb") as f:
        f.write(b"Pf\n" + f"{w} {h}\n".encode() + b"-1.0\n")
        f.write(np.flipud(data).astype("<f4").tobytes())


def test_kitti2015_lists_split_file_and_shuffle(tmp_path):
    root = str(tmp_path) + "/"
    for i in range(6):
        for name in (f"{i:06d}_10.png", f"{i:06d}_11.png"):
            _png(root + "image_2/" + name, np.zeros((2, 2, 3), np.uint8))
    split = tmp_path / "val.txt"
    split.write_text("4\n1\n")
    ltr, rtr, dtr, lva, rva, dva = D.kitti2015_lists(root, str(split))
    assert [os.path.basename(p) for p in lva] == ["000001_10.png", "000004_10.png"]          # sorted split
    assert sorted(os.path.basename(p) for p in ltr) == [f"{i:06d}_10.png" for i in (0, 2, 3, 5)]   # no _11 frames
    assert all("image_3/" in p for p in rtr + rva) and all("disp_occ_0/" in p for p in dtr + dva)
    assert [os.path.basename(p) for p in rva] == [os.path.basename(p) for p in lva]
    # no split file: first 40 of a shuffled arange(200)
    seen = {}

    def fake_shuffle(a):
        a[:] = a[::-1]
        seen["n"] = len(a)

    out = D.kitti2015_lists(root, None, shuffle=fake_shuffle)
    assert seen["n"] == 200 and os.path.basename(out[3][0]) == "000199_10.png" and len(out[3]) == 40


def test_sceneflow_lists_layout_and_quirks(tmp_path):
    root = str(tmp_path)
    img = np.zeros((2, 2, 3), np.uint8)

    def pair(base_img, base_disp, name="0006.png"):
        _png(f"{base_img}/left/{name}", img)
        _png(f"{base_img}/right/{name}", img)
        _pfm(f"{base_disp}/left/{name.split('.')[0]}.pfm", np.ones((2, 2), np.float32))

    pair(f"{root}/monkaa_frames_cleanpass/scene_a", f"{root}/monkaa_disparity/scene_a")
    for split in ("TRAIN", "TEST"):
        for part in ("A", "B", "C"):
            pair(f"{root}/frames_cleanpass/{split}/{part}/0001", f"{root}/frames_disparity/{split}/{part}/0001")
    for focal in ("15mm_focallength", "35mm_focallength"):
        for direction in ("scene_backwards", "scene_forwards"):
            for speed in ("fast", "slow"):
                pair(f"{root}/driving_frames_cleanpass/{focal}/{direction}/{speed}",
                     f"{root}/driving_disparity/{focal}/{direction}/{speed}")
    trl, trr, trd, tel, ter, ted = D.sceneflow_lists(root)
    assert len(tel) == len(ter) == len(ted) == 3 and all("/TEST/" in p for p in tel + ted)
    # monkaa 1 + flying TRAIN 3 + driving: 15 mm listed TWICE (4 folders x 2), 35 mm never
    assert len(trl) == len(trr) == len(trd) == 1 + 3 + 8
    assert sum("15mm_focallength" in p for p in trl) == 8 and not any("35mm" in p for p in trl)
    assert all(p.endswith(".pfm") for p in trd) and all(os.path.isfile(p) for p in trl + trr + trd)


def test_stereo_pairs_crop_rules(tmp_path):
    rng = np.random.default_rng(0)
    H, W = 375, 1242
    left = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    right = (rng.random((H, W, 3)) * 255).astype(np.uint8)
    disp16 = (rng.random((H, W)) * 60000).astype(np.uint16)
    _png(str(tmp_path / "l.png"), left)
    _png(str(tmp_path / "r.png"), right)
    Image.fromarray(disp16).save(tmp_path / "d.png")
    args = ([str(tmp_path / "l.png")], [str(tmp_path / "r.png")], [str(tmp_path / "d.png")])
    # KITTI evaluation: bottom-right 368x1232 of image AND ground truth, disparity = png / 256
    l, r, d = D.StereoPairs(*args, training=False, kitti_set=True)[0]
    assert l.shape == (3, 368, 1232) and d.shape == (368, 1232) and l.dtype == np.float32
    want = ((left[H - 368:, W - 1232:].astype(np.float32) / 255 - IMAGENET_MEAN) / IMAGENET_STD).transpose(2, 0, 1)
    assert np.allclose(l, want, atol=1e-6) and np.array_equal(d, disp16[H - 368:, W - 1232:].astype(np.float32) / 256)
    # training: random 256x512 window, same window for both images and the ground truth
    ds = D.StereoPairs(*args, training=True, kitti_set=True, rng=random.Random(3))
    l, r, d = ds[0]
    chk = random.Random(3)
    x1, y1 = chk.randint(0, W - 512), chk.randint(0, H - 256)
    assert l.shape == (3, 256, 512) and d.shape == (256, 512)
    assert np.array_equal(d, disp16[y1:y1 + 256, x1:x1 + 512].astype(np.float32) / 256)
    want = ((right[y1:y1 + 256, x1:x1 + 512].astype(np.float32) / 255 - IMAGENET_MEAN) / IMAGENET_STD).transpose(2, 0, 1)
    assert np.allclose(r, want, atol=1e-6)


def test_stereo_pairs_sceneflow_eval_pads_four_rows(tmp_path):
    rng = np.random.default_rng(1)
    left = (rng.random((540, 960, 3)) * 255).astype(np.uint8)
    _png(str(tmp_path / "l.png"), left)
    _png(str(tmp_path / "r.png"), left)
    gt = rng.random((540, 960)).astype(np.float32) * 100
    _pfm(str(tmp_path / "d.pfm"), gt)
    l, r, d = D.StereoPairs([str(tmp_path / "l.png")], [str(tmp_path / "r.png")], [str(tmp_path / "d.pfm")],
                            training=False, kitti_set=False)[0]
    assert l.shape == (3, 544, 960) and d.shape == (540, 960) and np.array_equal(d, gt)
    zero = ((0.0 - IMAGENET_MEAN) / IMAGENET_STD).astype(np.float32)
    assert np.allclose(l[:, :4, :], zero[:, None, None]) and np.allclose(
        l[:, 4:, :], ((left.astype(np.float32) / 255 - IMAGENET_MEAN) / IMAGENET_STD).transpose(2, 0, 1), atol=1e-6)
    from lwsnet_amd.synth import check_size
    check_size(544, 960, 32)                                     # BASELINE config 5's geometry is legal
