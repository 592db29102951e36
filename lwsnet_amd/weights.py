"""State-dict contract of the reference model and a seeded weight generator.

The key names and shapes follow what PaddlePaddle derives from the layer
definitions of the reference (``nn.Sequential`` / ``nn.LayerList`` children are
named by index, BatchNorm contributes ``weight bias _mean _variance``, every
convolution is ``bias_attr=False``):

* feature extractor   -- /root/reference/models/submodules.py:113-188 (+ hourglass :35-109)
* 3D filtering stacks -- /root/reference/models/submodules.py:190-221, models/models.py:19-22
* refinement          -- /root/reference/models/submodules.py:223-327, models/models.py:24-26

No trained checkpoint ships with the reference (README.md:122-123 are Drive
links), so benchmarks and tests use ``make_state_dict(seed)``: Kaiming-normal
convolutions (the initialiser the reference constructors request) and
*randomised* BatchNorm statistics so that BN is not an identity.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np

BN_SUFFIXES = ("weight", "bias", "_mean", "_variance")


def default_args(maxdisplist=(24, 5, 5), layers_3d=4, channels_3d=8, growth_rate=(4, 1, 1), feature_fp16=False,
                 interp_align_mode=0):
    """The namespace the reference CLI builds (inference.py:23-26); feature_fp16 is this build's BASELINE-config-5 switch,
    interp_align_mode its reading of Paddle's F.interpolate default (include/lwsnet_hip.h, lws_config)."""
    return SimpleNamespace(maxdisplist=list(maxdisplist), layers_3d=int(layers_3d),
                           channels_3d=int(channels_3d), growth_rate=list(growth_rate), feature_fp16=bool(feature_fp16),
                           interp_align_mode=int(interp_align_mode))


def _bn(spec, prefix, c):
    for s in BN_SUFFIXES:
        spec.append((f"{prefix}.{s}", (c,), "bn" + s))


def _conv(spec, key, shape):
    spec.append((key, tuple(shape), "conv"))


def state_dict_spec(args=None):
    """Ordered list of ``(key, shape, kind)`` for every tensor of the model."""
    args = args or default_args()
    spec = []
    fe = "feature_extraction"
    # dres0 / dres1: Sequential(convbn, ReLU, convbn[, ReLU])  (submodules.py:118-153)
    for blk, chans in (("dres0", ((4, 3), (8, 4))), ("dres1", ((4, 8), (8, 4)))):
        for idx, (co, ci) in zip((0, 2), chans):
            _conv(spec, f"{fe}.{blk}.{idx}.0.weight", (co, ci, 3, 3))
            _bn(spec, f"{fe}.{blk}.{idx}.1", co)
    # hourglass (submodules.py:40-92); conv1..4 are Sequential(convbn, ReLU)
    for name, (co, ci) in (("conv1", (16, 8)), ("conv2", (16, 16)), ("conv3", (16, 16)), ("conv4", (16, 16))):
        _conv(spec, f"{fe}.dres2.{name}.0.0.weight", (co, ci, 3, 3))
        _bn(spec, f"{fe}.dres2.{name}.0.1", co)
    # conv5/conv6 are deconvbn: Conv2DTranspose weight is [Cin, Cout, k, k]
    _conv(spec, f"{fe}.dres2.conv5.0.weight", (16, 16, 3, 3))
    _bn(spec, f"{fe}.dres2.conv5.1", 16)
    _conv(spec, f"{fe}.dres2.conv6.0.weight", (16, 8, 3, 3))
    _bn(spec, f"{fe}.dres2.conv6.1", 8)
    # classif1 (submodules.py:157-172)
    _conv(spec, f"{fe}.classif1.0.0.weight", (8, 8, 3, 3))
    _bn(spec, f"{fe}.classif1.0.1", 8)
    _conv(spec, f"{fe}.classif1.2.weight", (8, 8, 3, 3))
    # 3D stacks (submodules.py:216-221; models.py:19-22)
    for i in range(3):
        c3 = args.channels_3d * args.growth_rate[i]
        chans = [(1, c3)] + [(c3, c3)] * args.layers_3d + [(c3, 1)]
        for j, (ci, co) in enumerate(chans):
            _bn(spec, f"volume_postprocess.{i}.{j}.0", ci)
            _conv(spec, f"volume_postprocess.{i}.{j}.2.weight", (co, ci, 3, 3, 3))
    # refinement1_{left,disp} (submodules.py:282-300)
    for name, cin in (("refinement1_left", 3), ("refinement1_disp", 1)):
        _conv(spec, f"{name}.0.weight", (32, cin, 3, 3))
        for k in range(1, 5):
            _bn(spec, f"{name}.{k}.0", 32)
            _conv(spec, f"{name}.{k}.2.weight", (32, 1, 3, 3))
            _conv(spec, f"{name}.{k}.3.weight", (32, 32, 1, 1))
    # refinement2 (submodules.py:302-327)
    _bn(spec, "refinement2.0.0", 64)
    _conv(spec, "refinement2.0.2.weight", (32, 64, 3, 3))
    for k in range(1, 5):
        _bn(spec, f"refinement2.{k}.0", 32)
        _conv(spec, f"refinement2.{k}.2.weight", (32, 1, 3, 3))
        _conv(spec, f"refinement2.{k}.3.weight", (32, 32, 1, 1))
    _conv(spec, "refinement2.5.weight", (1, 32, 3, 3))
    return spec


def _fan_in(key, shape):
    # Paddle's KaimingNormal uses fan_in = shape[1] * receptive field for both
    # Conv and ConvTranspose weight tensors.
    rf = int(np.prod(shape[2:]))
    return shape[1] * rf


def make_state_dict(seed=7, args=None, calibrated=True):
    """Seeded synthetic weights: ``{key: float32 ndarray}`` in spec order.

    With ``calibrated`` (default) the BatchNorm running statistics are replaced
    by the ones in ``data/bn_calib_seed<seed>.npz`` (written by
    ``tools/make_bn_calibration.py``) when that file exists, so activations keep
    the O(1) scale of a trained network instead of growing layer by layer.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape, kind in state_dict_spec(args):
        if kind == "conv":
            std = np.sqrt(2.0 / _fan_in(key, shape))
            v = rng.standard_normal(shape) * std
        elif kind == "bnweight":
            v = rng.uniform(0.5, 1.5, shape)
        elif kind == "bnbias":
            v = rng.standard_normal(shape) * 0.1
        elif kind == "bn_mean":
            v = rng.standard_normal(shape) * 0.1
        elif kind == "bn_variance":
            v = rng.uniform(0.5, 1.5, shape)
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    if calibrated:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"bn_calib_seed{seed}.npz")
        if os.path.isfile(path):
            with np.load(path) as z:
                for key in z.files:
                    if key in sd and sd[key].shape == z[key].shape:
                        sd[key] = np.ascontiguousarray(z[key], dtype=np.float32)
    return sd
