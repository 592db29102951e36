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
