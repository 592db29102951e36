"""A torch-CPU stand-in for the ~25 PaddlePaddle calls /root/reference/models/{models,submodules}.py make.

BUILD-CONTAINER TOOLING ONLY (tools/check_oracle_vs_reference.py).  PaddlePaddle 2.0.0rc0 cannot be installed here
(no wheel, no network), so the reference's own source cannot run as published.  `install()` registers this module as
`paddle`, `paddle.nn`, `paddle.nn.functional`, `paddle.nn.initializer` in sys.modules so that the reference's source
files can be imported IN PLACE from /root/reference and executed line by line.

What this does and does not pin:
* it removes transcription risk from oracle/lws_oracle.py: control flow, layer wiring, state-dict names, padding rules,
  index arithmetic of the reference are executed from the reference's own text;
* it pins NOTHING about Paddle's kernels or op defaults.  Every default below (grid_sample align_corners=True,
  interpolate align_corners=False/align_mode=0, `tensor / python_scalar` lowered to a multiply by the float32
  reciprocal, BatchNorm epsilon 1e-5) is the same from-memory reading of Paddle 2.0 that SURVEY.md appendix B lists.
  The oracle therefore stays "parity unpinned".

Nothing in the product, the tests' run-time path, bench.py or smoke() imports this file; /root/reference never
travels to the GPU box.
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn.functional as TF

_DTYPE = torch.float32        # what the reference's hard-coded dtype='float32' maps to (float64 for the "truth" run)


def set_dtype(dt):
    global _DTYPE
    _DTYPE = dt


def _dt(name):
    if name in (None, "float32"):
        return _DTYPE
    if name == "float64":
        return torch.float64
    if name == "int64":
        return torch.int64
    raise NotImplementedError(name)


def _is_scalar(v):
    return isinstance(v, (int, float)) and not isinstance(v, bool)


class Tensor(torch.Tensor):
    """torch.Tensor with the Paddle method spellings the reference uses.  Python-scalar arithmetic follows Paddle's
    dygraph math-op patch: `x * c`, `x / c`, `x - c` become one `scale` op, and `x / c`
    multiplies by the reciprocal rounded to the tensor dtype."""

    @staticmethod
    def wrap(t):
        return t.as_subclass(Tensor) if isinstance(t, torch.Tensor) and not isinstance(t, Tensor) else t

    # -- scalar arithmetic --------------------------------------------------------------------
    def __truediv__(self, other):
        if _is_scalar(other):
            from oracle import lws_oracle as O
            if O.VARIANT["scalar_div"] == "divide":              # the other reading: a true elementwise division
                return torch.Tensor.__truediv__(self, torch.tensor(float(other), dtype=self.dtype))
            r = torch.tensor(1.0, dtype=self.dtype) / torch.tensor(float(other), dtype=self.dtype)
            return torch.Tensor.__mul__(self, r)
        return torch.Tensor.__truediv__(self, other)

    def __mul__(self, other):
        if _is_scalar(other):
            return torch.Tensor.__mul__(self, torch.tensor(float(other), dtype=self.dtype))
        return torch.Tensor.__mul__(self, other)

    __rmul__ = __mul__

    # -- method spellings ---------------------------------------------------------------------
    def reshape(self, *a, shape=None):
        return torch.Tensor.reshape(self, *(a if shape is None else (list(shape),)))

    def expand(self, *a, shape=None):
        return torch.Tensor.expand(self, *(a if shape is None else tuple(shape)))

    def unsqueeze(self, axis):
        t = self
        for ax in (axis if isinstance(axis, (list, tuple)) else [axis]):
            t = torch.Tensor.unsqueeze(t, ax)
        return t

    def squeeze(self, axis=None):
        if axis is None:
            return torch.Tensor.squeeze(self)
        t = self
        for ax in sorted(axis if isinstance(axis, (list, tuple)) else [axis], reverse=True):
            t = torch.Tensor.squeeze(t, ax)
        return t

    def sum(self, axis=None, keepdim=False):
        return torch.Tensor.sum(self) if axis is None else torch.Tensor.sum(self, dim=axis, keepdim=keepdim)

    def numpy(self):
        return torch.Tensor.numpy(self.detach().as_subclass(torch.Tensor))

    @property
    def stop_gradient(self):
        return not self.requires_grad

    @stop_gradient.setter
    def stop_gradient(self, v):      # the reference toggles this on leaf tensors; inference only, so it is inert here
        pass


def to_tensor(a, dtype=None):
    return Tensor.wrap(torch.as_tensor(a, dtype=_dt(dtype) if dtype is not None or not isinstance(a, torch.Tensor) else None))


# ---- paddle.* ----------------------------------------------------------------------------------
def arange(start, end=None, step=1, dtype=None):
    if end is None:
        start, end = 0, start
    return Tensor.wrap(torch.arange(start, end, step, dtype=_dt(dtype)))


def expand(x, shape):
    return Tensor.wrap(x).expand(shape=shape)


def reshape(x, shape):
    return Tensor.wrap(x).reshape(shape=shape)


def concat(xs, axis=0):
    return Tensor.wrap(torch.cat(list(xs), dim=axis))


def transpose(x, perm):
    return Tensor.wrap(x.permute(*perm))


def zeros(shape, dtype=None):
    return Tensor.wrap(torch.zeros(tuple(shape), dtype=_dt(dtype)))


def norm(x, p, axis):
    assert p == 1, "the reference only takes L1 norms (models/models.py:72,74,101)"
    return Tensor.wrap(x.abs().sum(axis=axis))


def unsqueeze(x, axis):
    return Tensor.wrap(x).unsqueeze(axis)


def squeeze(x, axis=None):
    return Tensor.wrap(x).squeeze(axis)


def sum(x, axis=None, keepdim=False):     # noqa: A001  (paddle.sum)
    return Tensor.wrap(x).sum(axis=axis, keepdim=keepdim)


class no_grad:
    def __enter__(self):
        self._g = torch.no_grad()
        return self._g.__enter__()

    def __exit__(self, *a):
        return self._g.__exit__(*a)


# ---- paddle.nn ---------------------------------------------------------------------------------
class Layer(torch.nn.Module):
    def set_state_dict(self, sd):
        own = dict(self.state_dict())
        missing = [k for k in own if k not in sd]
        extra = [k for k in sd if k not in own and k != "StructuredToParameterName@@"]
        if missing or extra:
            raise KeyError(f"state dict mismatch: missing {missing[:4]} ({len(missing)}), unexpected {extra[:4]} ({len(extra)})")
        with torch.no_grad():
            for k, v in own.items():
                src = torch.as_tensor(sd[k])
                if tuple(src.shape) != tuple(v.shape):
                    raise ValueError(f"{k}: shape {tuple(src.shape)} != {tuple(v.shape)}")
                v.copy_(src.to(v.dtype))

    def __call__(self, *a, **kw):
        a = tuple(Tensor.wrap(x) for x in a)
        kw = {k: Tensor.wrap(v) for k, v in kw.items()}
        return super().__call__(*a, **kw)


class Sequential(torch.nn.Sequential, Layer):
    pass


class LayerList(torch.nn.ModuleList, Layer):
    pass


class ReLU(Layer):
    def forward(self, x):
        return Tensor.wrap(TF.relu(x))


def _tup(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


class _ConvNd(Layer):
    ND = 2

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 weight_attr=None, bias_attr=None):
        super().__init__()
        assert bias_attr is False, "every convolution of the reference is bias_attr=False"
        n = self.ND
        self.stride, self.padding, self.dilation, self.groups = _tup(stride, n), _tup(padding, n), _tup(dilation, n), groups
        self.weight = torch.nn.Parameter(torch.zeros((out_channels, in_channels // groups) + _tup(kernel_size, n), dtype=_DTYPE),
                                         requires_grad=False)

    def forward(self, x):
        f = TF.conv2d if self.ND == 2 else TF.conv3d
        return Tensor.wrap(f(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups))


class Conv2D(_ConvNd):
    ND = 2


class Conv3D(_ConvNd):
    ND = 3


class Conv2DTranspose(Layer):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0,
                 weight_attr=None, bias_attr=None):
        super().__init__()
        assert bias_attr is False
        self.stride, self.padding, self.output_padding = stride, padding, output_padding
        self.weight = torch.nn.Parameter(torch.zeros((in_channels, out_channels) + _tup(kernel_size, 2), dtype=_DTYPE),
                                         requires_grad=False)

    def forward(self, x):
        return Tensor.wrap(TF.conv_transpose2d(x, self.weight, None, self.stride, self.padding, self.output_padding))


class _BatchNorm(Layer):
    """Eval-mode only; state keys weight / bias / _mean / _variance, epsilon 1e-5 (Paddle's names and default)."""

    def __init__(self, num_features, momentum=0.9, epsilon=1e-5):
        super().__init__()
        self.epsilon = epsilon
        self.weight = torch.nn.Parameter(torch.ones(num_features, dtype=_DTYPE), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.zeros(num_features, dtype=_DTYPE), requires_grad=False)
        self.register_buffer("_mean", torch.zeros(num_features, dtype=_DTYPE))
        self.register_buffer("_variance", torch.ones(num_features, dtype=_DTYPE))

    def forward(self, x):
        assert not self.training, "the stand-in implements inference (running statistics) only"
        return Tensor.wrap(TF.batch_norm(x, self._mean, self._variance, self.weight, self.bias, False, 0.0, self.epsilon))


class BatchNorm2D(_BatchNorm):
    pass


class BatchNorm3D(_BatchNorm):
    pass


class KaimingNormal:
    pass


# ---- paddle.nn.functional ------------------------------------------------------------------------
def relu(x):
    return Tensor.wrap(TF.relu(x))


def softmax(x, axis=-1):
    return Tensor.wrap(TF.softmax(x, dim=axis))


# The stand-in's readings of Paddle's defaults are the SAME switches as the restatement's (oracle.lws_oracle.VARIANT:
# align_mode, grid_unnorm, scalar_div) and share its implementations, so tools/oracle_sensitivity.py can run the reference's
# own source under each reading.
def interpolate(x, size=None, mode="nearest", align_corners=False, align_mode=None):
    from oracle import lws_oracle as O
    assert mode == "bilinear" and align_corners is False
    am = O.VARIANT["align_mode"] if align_mode is None else align_mode      # (the reference never passes align_mode)
    return Tensor.wrap(O.interp_bilinear(x, list(size), am))


def grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True):
    from oracle import lws_oracle as O
    assert mode == "bilinear" and padding_mode == "zeros" and align_corners is True
    return Tensor.wrap(O.grid_sample_bilinear(x, grid, O.VARIANT["grid_unnorm"]))


def install():
    """Registers the stand-in as `paddle` (+ submodules) in sys.modules.  Returns the top-level module."""
    if "paddle" in sys.modules and not getattr(sys.modules["paddle"], "_lws_shim", False):
        raise RuntimeError("a real `paddle` is importable: use it instead of the stand-in")
    me = sys.modules[__name__]
    top = types.ModuleType("paddle")
    top._lws_shim = True
    for name in ("arange", "expand", "reshape", "concat", "transpose", "zeros", "norm", "unsqueeze", "squeeze", "sum",
                 "no_grad", "to_tensor", "Tensor"):
        setattr(top, name, getattr(me, name))
    nn = types.ModuleType("paddle.nn")
    for name in ("Layer", "Sequential", "LayerList", "ReLU", "Conv2D", "Conv3D", "Conv2DTranspose", "BatchNorm2D",
                 "BatchNorm3D"):
        setattr(nn, name, getattr(me, name))
    init = types.ModuleType("paddle.nn.initializer")
    init.KaimingNormal = KaimingNormal
    fn = types.ModuleType("paddle.nn.functional")
    for name in ("relu", "softmax", "interpolate", "grid_sample"):
        setattr(fn, name, getattr(me, name))
    nn.initializer = init
    nn.functional = fn
    top.nn = nn
    sys.modules.update({"paddle": top, "paddle.nn": nn, "paddle.nn.initializer": init, "paddle.nn.functional": fn})
    return top
