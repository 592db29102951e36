"""Host-side pieces around the path (CPU): checkpoint ingest, image I/O rules, metrics."""
import pickle

import numpy as np
import pytest

from lwsnet_amd import checkpoint, imageio, metrics
from lwsnet_amd.weights import make_state_dict


def test_pdparams_roundtrip_and_restricted_unpickler(tmp_path, state_dict):
    p = tmp_path / "ckpt.pdparams"
    checkpoint.save_pdparams(state_dict, p)
    sd = checkpoint.load_state_dict(p)
    assert set(sd) == set(state_dict)                       # bookkeeping entry dropped
    assert all(np.array_equal(sd[k], state_dict[k]) for k in sd)
    # paddle >= 2.1 layout: (name, ndarray) tuples
    with open(tmp_path / "v21.pdparams", "wb") as f:
        pickle.dump({k: (k, v) for k, v in list(state_dict.items())[:5]}, f, protocol=4)
    sd21 = checkpoint.load_state_dict(tmp_path / "v21.pdparams")
    assert len(sd21) == 5 and all(isinstance(v, np.ndarray) for v in sd21.values())
    # a pickle that references anything but numpy is rejected, not executed
    with open(tmp_path / "evil.pdparams", "wb") as f:
        pickle.dump({"x": print}, f)
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.load_state_dict(tmp_path / "evil.pdparams")
    checkpoint.save_npz(state_dict, tmp_path / "w.npz")
    assert np.array_equal(checkpoint.load_state_dict(tmp_path / "w.npz")["refinement2.5.weight"],
                          state_dict["refinement2.5.weight"])


def _paddle_names(state_dict):
    """Paddle's internal parameter names for the reference's layer tree, as dygraph's unique-name generator hands them out
    in construction order: conv2d_N.w_0, batch_norm_N.w_0 / .b_0 / .w_1 (_mean) / .w_2 (_variance), conv3d_N.w_0, ..."""
    counters, owner, table = {}, {}, {}
    for key in state_dict:
        layer, leaf = key.rsplit(".", 1)
        if layer not in owner:
            v = np.asarray(state_dict[key])
            kind = ("batch_norm" if leaf in ("weight", "bias", "_mean", "_variance") and v.ndim == 1 else
                    "conv3d" if v.ndim == 5 else "conv2d_transpose" if ".conv5." in key or ".conv6." in key else "conv2d")
            owner[layer] = f"{kind}_{counters.setdefault(kind, 0)}"
            counters[kind] += 1
        table[key] = owner[layer] + {"weight": ".w_0", "bias": ".b_0", "_mean": ".w_1", "_variance": ".w_2"}[leaf]
    return table


def test_pdparams_as_paddle_2_0rc0_writes_it(tmp_path, state_dict):
    """`paddle.save(model.state_dict(), path)` in 2.0.0rc0 (train.py:115): `_build_saved_state_dict` turns every VarBase into
    `value.numpy()`, collects `{structured name: value.name}` under "StructuredToParameterName@@" and pickles the dict with
    protocol 2.  Built here by hand (not with checkpoint.save_pdparams), including the internal parameter names, an
    OrderedDict container and non-contiguous / float64 arrays a user-side conversion might leave behind."""
    import collections
    saved = collections.OrderedDict()
    for i, (k, v) in enumerate(state_dict.items()):
        a = np.asarray(v)
        saved[k] = np.asfortranarray(a) if i % 7 == 0 else (a.astype(np.float64) if i % 11 == 0 else a)
    saved["StructuredToParameterName@@"] = _paddle_names(state_dict)
    with open(tmp_path / "rc0.pdparams", "wb") as f:
        pickle.dump(saved, f, protocol=2)
    sd = checkpoint.load_state_dict(tmp_path / "rc0.pdparams")
    assert list(sd) == list(state_dict) and len(sd) == 226
    for k in sd:
        assert sd[k].dtype == np.float32 and sd[k].flags["C_CONTIGUOUS"] and np.array_equal(sd[k], state_dict[k]), k
    # 2.0 final writes an (empty) big-parameter table when nothing was split; a non-empty one is refused, not mis-read
    saved["UnpackBigParamInfor@@"] = {}
    with open(tmp_path / "v20.pdparams", "wb") as f:
        pickle.dump(dict(saved), f, protocol=2)
    assert set(checkpoint.load_state_dict(tmp_path / "v20.pdparams")) == set(state_dict)
    saved["UnpackBigParamInfor@@"] = {"refinement2.5.weight": {"OriginShape": (1, 32, 3, 3), "slices": ["a", "b"]}}
    with open(tmp_path / "split.pdparams", "wb") as f:
        pickle.dump(dict(saved), f, protocol=2)
    with pytest.raises(ValueError, match="split"):
        checkpoint.load_state_dict(tmp_path / "split.pdparams")


def test_pdparams_as_paddle_2_1_writes_it(tmp_path, state_dict):
    """Paddle >= 2.1 (`paddle.save` -> `_pickle_save` with `reduce_varbase`): each tensor is pickled as the tuple
    (internal parameter name, ndarray), protocol 4 by default, and there is no name table."""
    names = _paddle_names(state_dict)
    saved = {k: (names[k], np.asarray(v)) for k, v in state_dict.items()}
    for proto in (2, 4):
        p = tmp_path / f"v21_p{proto}.pdparams"
        with open(p, "wb") as f:
            pickle.dump(saved, f, protocol=proto)
        sd = checkpoint.load_state_dict(p)
        assert list(sd) == list(state_dict)
        assert all(np.array_equal(sd[k], state_dict[k]) and sd[k].dtype == np.float32 for k in sd)
    # integer arrays, non-string keys and objects are refused with a ValueError / UnpicklingError, never executed
    for bad, exc in (({"w": np.arange(4)}, ValueError), ({3: np.zeros(2, np.float32)}, ValueError),
                     ({"w": ("n", [1.0, 2.0])}, ValueError), ({"w": np.zeros(2, np.float32), "x@@": 3}, ValueError)):
        with open(tmp_path / "bad.pdparams", "wb") as f:
            pickle.dump(bad, f, protocol=2)
        with pytest.raises(exc):
            checkpoint.load_state_dict(tmp_path / "bad.pdparams")


def test_pdparams_loads_into_the_model_spec(tmp_path, state_dict):
    """A checkpoint in the 2.0rc0 layout passes the library's own key / shape check (lws_set_tensor's spec is derived from
    the constructor arguments exactly as the reference's layer tree is): every one of the 226 tensors is accepted."""
    import collections
    from lwsnet_amd.weights import default_args, state_dict_spec
    checkpoint.save_pdparams(state_dict, tmp_path / "m.pdparams")
    sd = checkpoint.load_state_dict(tmp_path / "m.pdparams")
    spec = {k: shape for k, shape, _ in state_dict_spec(default_args())}
    assert sorted(sd) == sorted(spec) and len(spec) == 226
    assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)


def test_crop_and_normalise_follow_inference_py():
    img = np.arange(375 * 1242 * 3, dtype=np.uint32).reshape(375, 1242, 3).astype(np.uint8)
    c = imageio.crop_bottom_right(img)
    assert c.shape == (368, 1232, 3) and np.array_equal(c, img[7:, 10:])          # inference.py:99
    assert imageio.crop_bottom_right(img[:300]) is None                            # :96-97 skip
    x = imageio.to_input(c)
    assert x.shape == (3, 368, 1232) and x.dtype == np.float32
    np.testing.assert_allclose(x[1, 0, 0], (c[0, 0, 1] / 255.0 - 0.456) / 0.224, rtol=1e-6)


def test_colour_map_and_uint8_cast():
    lut = imageio.jet_lut()
    assert lut.shape == (256, 3) and tuple(lut[0]) == (0, 0, 128) and tuple(lut[255]) == (128, 0, 0)
    d = np.array([[0.0, 95.9, 255.0, 256.0]])
    col = imageio.disparity_to_color(d)
    assert col.shape == (1, 4, 3)
    assert np.array_equal(col[0, 1], lut[95]) and np.array_equal(col[0, 3], lut[0])   # truncation, wrap at 256


def test_png_writer_roundtrips_through_pil(tmp_path):
    """imageio.encode_png / save_png (the CLI's cv2.imwrite, inference.py:120,136): a valid 8-bit RGB PNG that PIL decodes back
    to the same pixels -- random bytes, a smooth colour-mapped disparity, one pixel, odd sizes."""
    from PIL import Image
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:41, 0:77]
    cases = [rng.integers(0, 256, (368, 1232, 3), dtype=np.uint8), rng.integers(0, 256, (1, 1, 3), dtype=np.uint8),
             rng.integers(0, 256, (7, 5, 3), dtype=np.uint8), imageio.disparity_to_color((3.0 * yy + 0.5 * xx).astype(np.float32))]
    for i, a in enumerate(cases):
        path = tmp_path / f"{i}.png"
        imageio.save_png(str(path), a)
        with Image.open(path) as im:
            assert im.mode == "RGB" and im.size == (a.shape[1], a.shape[0])
            assert np.array_equal(np.asarray(im), a)
    with pytest.raises(ValueError):
        imageio.encode_png(np.zeros((4, 4), np.uint8))


def test_jet_table_is_opencvs_against_the_reference_pngs():
    """The reference's own reference/{1..4}.png (written by inference.py:114-120) are the golden: every pixel is one
    entry of cv2.COLORMAP_JET.  Fixture = their distinct colours + four raw crops (tools/make_jet_fixture.py)."""
    from conftest import golden
    fx = golden("jet_reference_colours.npz")
    lut = imageio.jet_lut()
    assert len(np.unique(lut, axis=0)) == 256                                     # injective: levels are decodable
    index = {tuple(c): i for i, c in enumerate(lut.tolist())}
    cols = [tuple(c) for c in fx["colours"].tolist()]
    assert len(cols) == 128 and fx["counts"].sum() == 4 * 368 * 1232
    assert all(c in index for c in cols), [c for c in cols if c not in index]     # every reference colour is an entry
    levels = sorted(index[c] for c in cols)
    assert levels == list(range(128))                                             # a gap-free ramp from level 0
    # ramp order is the physical JET order, checked without the table: blue rises, then green, then red rises / blue falls
    by_level = [lut[i].astype(int) for i in range(128)]
    for a, b in zip(by_level, by_level[1:]):
        assert (b - a).tolist() in ([0, 0, 4], [0, 0, 3], [0, 4, 0], [0, 3, 0], [2, 3, -1], [4, 0, -4]), (a, b)
    # structure of the whole table: slope-4 piecewise-linear ramps with plateaux, endpoints, mirror symmetry
    assert np.array_equal(lut[::-1, ::-1], lut)
    steps = np.abs(np.diff(lut.astype(int), axis=0))
    assert set(np.unique(steps).tolist()) <= {0, 1, 2, 3, 4} and (steps.sum(1) > 0).all()
    assert [tuple(lut[i]) for i in (32, 96, 159, 160, 223, 224)] == \
        [(0, 0, 255), (2, 255, 254), (254, 255, 2), (255, 252, 0), (255, 0, 0), (252, 0, 0)]
    # raw crops of the four images: decode to levels and re-encode -> the same bytes
    patch = fx["patch"]
    lv = np.array([index[tuple(px)] for px in patch.reshape(-1, 3).tolist()], dtype=np.float32).reshape(patch.shape[:-1])
    assert np.array_equal(imageio.disparity_to_color(lv), patch)


def test_pfm_reader(tmp_path):
    data = np.arange(12, dtype="<f4").reshape(3, 4)
    with open(tmp_path / "d.pfm", "wb") as f:
        f.write(b"Pf\n4 3\n-1.0\n")
        f.write(np.flipud(data).tobytes())
    got, scale = imageio.read_pfm(tmp_path / "d.pfm")
    assert scale == 1.0 and np.array_equal(got, data)


def test_metrics_match_reference_formulas():
    gt = np.array([[10.0, 100.0, 0.0, 250.0, 50.0]])
    d = np.array([[14.0, 104.0, 5.0, 0.0, 52.0]])
    assert metrics.error_3px(d, gt) == pytest.approx(1.0 / 3.0)
    assert metrics.end_point_error(d, gt) == pytest.approx((4 + 4 + 5 + 2) / 4.0)


# ---- drop-in boundary: the expressions /root/reference/inference.py:102-103,108,114 evaluate -----------------------
class FakePaddleTensor:
    """Duck type of a Paddle 2.0 tensor as the reference's caller builds it (paddle.vision ToTensor/Normalize output):
    `.shape` is a list, `.unsqueeze(axis=0)`, `.numpy()`; it is neither a numpy array nor a torch tensor."""

    def __init__(self, a):
        self._a = np.asarray(a, dtype=np.float32)

    @property
    def shape(self):
        return list(self._a.shape)

    def unsqueeze(self, axis):
        return FakePaddleTensor(np.expand_dims(self._a, axis))

    def numpy(self):
        return self._a


def test_inputs_accept_paddle_like_tensors():
    import torch
    from lwsnet_amd.models import as_input
    img = np.random.default_rng(0).standard_normal((3, 16, 32)).astype(np.float32)
    left_input = FakePaddleTensor(img).unsqueeze(axis=0)                     # inference.py:102
    t = as_input(left_input, "left_input")
    assert isinstance(t, torch.Tensor) and tuple(t.shape) == (1, 3, 16, 32) and t.dtype == torch.float32
    assert np.array_equal(t.numpy(), img[None])

    class ArrayOnly:                                                         # __array__ protocol only
        def __array__(self, dtype=None, copy=None):
            return img[None]

    assert np.array_equal(as_input(ArrayOnly(), "x").numpy(), img[None])
    assert np.array_equal(as_input(torch.from_numpy(img[None]).double(), "x").numpy(), img[None])
    with pytest.raises(TypeError):
        as_input("left.png", "left_input")
    with pytest.raises(ValueError):
        as_input(FakePaddleTensor(img), "left_input")                        # forgot the batch axis


def test_outputs_answer_the_callers_paddle_spellings():
    import torch
    from lwsnet_amd.models import DisparityTensor
    d = np.random.default_rng(1).random((1, 1, 6, 10)).astype(np.float32) * 190
    outputs = [DisparityTensor.wrap(torch.from_numpy(d.copy())) for _ in range(4)]
    for stage in range(4):
        outputs[stage] = outputs[stage].squeeze(axis=[0, 1]).numpy().astype(np.uint8)      # inference.py:114
        assert outputs[stage].shape == (6, 10) and np.array_equal(outputs[stage], d[0, 0].astype(np.uint8))
    o = DisparityTensor.wrap(torch.from_numpy(d.copy()))
    assert tuple(o.squeeze(1).shape) == (1, 6, 10)                            # paddle.squeeze(outputs[stage], 1) (train.py:188)
    assert tuple(o.squeeze().shape) == (6, 10) and tuple(o.squeeze(axis=0).unsqueeze(axis=[0, 1]).shape) == (1, 1, 1, 6, 10)
    assert o.shape[2] == 6 and list(o.shape) == [1, 1, 6, 10]
    assert np.array_equal(np.asarray(o), d) and np.array_equal(o[0, 0].numpy(), d[0, 0])
    assert isinstance(o + 1.0, torch.Tensor)                                  # still a torch tensor for everything else
    assert np.array_equal(torch.from_dlpack(o).numpy(), d)                    # zero-copy hand-over to other frameworks


def test_overlap_account_on_a_synthetic_trace(tmp_path):
    """tools/overlap_account.py (the evidence behind DESIGN.md's "co-resident kernels time-share the CUs" statement): on a
    synthetic kernel trace whose side kernel runs fully beside a chain kernel without slowing it, the tool must report both
    kernels at their alone durations and 2.00x inside the two-queue windows; with the chain kernel stretched by exactly the side
    kernel's length (pure time sharing) it must report 1.00x."""
    import csv
    import os
    import subprocess
    import sys
    from conftest import ROOT
    for stretch, want in ((0, "2.00x"), (10, "1.00x")):
        rows, t = [], 0
        for f in range(40):
            over = f % 2 == 0
            for name, dur in (("void lws::k_conv2d_pair<3, 4, 8, 2>(float)", 10), ("void lws::k_a(float)", 20), ("void lws::k_b(float)", 30)):
                d = dur + (stretch if (over and name.endswith("k_b(float)")) else 0)
                rows.append(dict(Kernel_Name=name, Queue_Id=1, Start_Timestamp=t, End_Timestamp=t + d * 1000, Grid_Size_X=1, Grid_Size_Y=1, Grid_Size_Z=1))
                if over and name.endswith("k_b(float)"):
                    # the side kernel alone takes 10 us; beside k_b it takes 10 (no interference) or 20 (time sharing)
                    sd = 10 if stretch == 0 else 20
                    rows.append(dict(Kernel_Name="void lws::k_side(float)", Queue_Id=2, Start_Timestamp=t + 5000, End_Timestamp=t + 5000 + sd * 1000,
                                     Grid_Size_X=1, Grid_Size_Y=1, Grid_Size_Z=1))
                t += d * 1000
            # an un-overlapped launch of the side kernel per forward, so that its alone duration is known
            rows.append(dict(Kernel_Name="void lws::k_side(float)", Queue_Id=2, Start_Timestamp=t, End_Timestamp=t + 10000, Grid_Size_X=1, Grid_Size_Y=1, Grid_Size_Z=1))
            t += 10000
        path = tmp_path / f"kt_{stretch}.csv"
        with open(path, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "overlap_account.py"), str(path), "20", "4"], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        assert want in p.stdout, p.stdout
