"""Drop-in mirror of the reference model object (/root/reference/models/models.py:7-164).

``LWSNet(args)`` reads the same four namespace fields, ``set_state_dict`` takes the
same structured keys, and ``model(left, right)`` returns the same list of four
``[B,1,H,W]`` float32 full-resolution disparity maps.  The whole forward -- feature
extractor, the three volume stages (models.py:115-156) and the refinement -- runs in
the hand-written HIP library behind the C ABI (``lws_forward``); this class only owns
device buffers.  There is no CPU path.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib, ops
from .synth import check_size
from .weights import state_dict_spec


class DisparityTensor(torch.Tensor):
    """What ``model(left, right)`` returns per stage: a device-resident ``torch.Tensor`` that also answers the Paddle
    spellings the reference's callers use on the outputs -- ``outputs[stage].squeeze(axis=[0, 1]).numpy()``
    (/root/reference/inference.py:114), ``.unsqueeze(axis=...)``, ``.numpy()`` on a device tensor (train.py:188).
    ``.numpy()`` is the device-to-host copy and therefore the synchronisation point, exactly as with a Paddle GPU tensor.
    Everything else is plain torch; ``__dlpack__`` (inherited) hands the buffer to any other framework without a copy."""

    @staticmethod
    def wrap(t):
        return t.as_subclass(DisparityTensor)

    def squeeze(self, dim=None, axis=None):
        ax = axis if axis is not None else dim
        if ax is None:
            return torch.Tensor.squeeze(self)
        t = self
        for a in sorted([int(v) % self.dim() for v in (ax if isinstance(ax, (list, tuple)) else [ax])], reverse=True):
            t = torch.Tensor.squeeze(t, a)
        return t

    def unsqueeze(self, dim=None, axis=None):
        ax = axis if axis is not None else dim
        t = self
        for a in (ax if isinstance(ax, (list, tuple)) else [ax]):
            t = torch.Tensor.unsqueeze(t, int(a))
        return t

    def numpy(self):
        return torch.Tensor.numpy(self.detach().as_subclass(torch.Tensor).cpu())

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)


def as_input(x, name, device=None):
    """Accepts what the reference's callers hand to ``model(left, right)``: a Paddle tensor is anything that exports
    DLPack (device buffers are taken without a copy) or answers ``.numpy()`` / ``__array__`` (inference.py:102-103
    builds its inputs with paddle.vision transforms + ``.unsqueeze(axis=0)``); numpy arrays and torch tensors too.
    Returns a contiguous float32 ``[B,3,H,W]`` torch tensor on `device`."""
    if isinstance(x, torch.Tensor):
        t = x
    elif isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    elif hasattr(x, "__dlpack__") and hasattr(x, "__dlpack_device__") and x.__dlpack_device__()[0] != 1:
        t = torch.from_dlpack(x)                       # a device tensor of another framework (kDLCPU == 1 goes below)
    elif callable(getattr(x, "numpy", None)):
        t = torch.from_numpy(np.ascontiguousarray(x.numpy(), dtype=np.float32))
    elif hasattr(x, "__array__"):
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32))
    else:
        raise TypeError(f"{name} must be a tensor (torch / DLPack / .numpy() / __array__) or a numpy array, got {type(x).__name__}")
    if t.dim() != 4 or t.shape[1] != 3:
        raise ValueError(f"{name} must be [B,3,H,W]; got {tuple(t.shape)}")
    return t.detach().as_subclass(torch.Tensor).to(device=device, dtype=torch.float32).contiguous()


class LWSNet:
    def __init__(self, args, device=None):
        self.maxdisplist = [int(v) for v in args.maxdisplist]      # models.py:11-14
        self.layers_3d = int(args.layers_3d)
        self.channels_3d = int(args.channels_3d)
        self.growth_rate = [int(v) for v in args.growth_rate]
        if len(self.maxdisplist) != 3 or len(self.growth_rate) != 3:
            raise ValueError("maxdisplist and growth_rate must have three entries (one per volume stage)")
        self._args = args
        self._spec = {k: s for k, s, _ in state_dict_spec(args)}
        self._sd = {}
        self._params = None
        self._training = False
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else None
        lib = _lib.load()                                         # raises if the HIP extension is missing
        self.feature_fp16 = bool(getattr(args, "feature_fp16", False))     # BASELINE config 5 (not in the reference)
        # which source index the reference's four F.interpolate calls use (models.py:119,146,154,161): 0 = half-pixel centres
        # (this build's reading of Paddle 2.0rc0), 1 = src = ratio * dst; not in the reference's namespace
        self.interp_align_mode = int(getattr(args, "interp_align_mode", 0))
        if self.interp_align_mode not in (0, 1):
            raise ValueError(f"interp_align_mode must be 0 or 1, got {self.interp_align_mode}")
        cfg = _lib.LwsConfig((ctypes.c_int32 * 3)(*self.maxdisplist), self.layers_3d, self.channels_3d,
                             (ctypes.c_int32 * 3)(*self.growth_rate), 1 if self.feature_fp16 else 0, self.interp_align_mode)
        self._h = ctypes.c_void_p()
        with self._device_ctx():
            _lib.check(lib.lws_create(ctypes.byref(cfg), ctypes.byref(self._h)), "lws_create")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.load().lws_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- nn.Layer protocol used by inference.py / train.py ------------------------------------
    def eval(self):
        self._training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("lwsnet_amd.LWSNet is inference-only: BatchNorm uses running statistics")
        return self.eval()

    def state_dict(self):
        return dict(self._sd)

    def parameters(self):
        return [v for k, v in self._sd.items() if not k.endswith(("._mean", "._variance"))]

    def set_state_dict(self, state_dict):
        """model.set_state_dict(paddle.load(path)) (inference.py:45): {structured name: array}."""
        sd = {}
        for k, v in state_dict.items():
            if k == "StructuredToParameterName@@":               # bookkeeping entry of paddle.save (2.0rc0)
                continue
            if isinstance(v, tuple) and len(v) == 2:               # paddle >= 2.1 stores (name, ndarray)
                v = v[1]
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            sd[k] = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
        missing = [k for k in self._spec if k not in sd]
        if missing:
            raise KeyError(f"state dict is missing {len(missing)} entries, e.g. {missing[:3]}")
        lib = _lib.load()
        with self._device_ctx():
            for k, v in sd.items():
                shape = (ctypes.c_int64 * v.ndim)(*v.shape)
                _lib.check(lib.lws_set_tensor(self._h, k.encode(), v.ctypes.data_as(_lib.c_float_p), shape, v.ndim),
                           "lws_set_tensor")
            if self.device is not None:
                _lib.check(lib.lws_finalize(self._h), "lws_finalize")
                self._params = True
        self._sd = sd
        return self

    load_dict = set_state_dict

    def set_option(self, name, value):
        """Launch-plan option of the HIP library (include/lwsnet_hip.h: lws_set_option); results never change."""
        _lib.check(_lib.load().lws_set_option(self._h, name.encode(), int(value)), "lws_set_option")
        return self

    def get_option(self, name):
        v = ctypes.c_int(0)
        _lib.check(_lib.load().lws_get_option(self._h, name.encode(), ctypes.byref(v)), "lws_get_option")
        return v.value

    # ---- several forwards in flight (include/lwsnet_hip.h: lws_pool_*) --------------------------
    def pool(self, workers=3, side_streams=False):
        """A pool of `workers` host threads inside the HIP library, each with its own clone of this model (shared
        parameters, private workspace) and one HIP stream: batch-1 forwards are launch-latency-bound chains of ~35
        dependent kernels, and keeping a few of them in flight overlaps them.  Bit-identical to forward()."""
        if self._params is None:
            raise RuntimeError("set_state_dict() must be called before pool()")
        return ForwardPool(self, int(workers), bool(side_streams))

    def map(self, pairs, workers=3):
        """`[model(l, r) for l, r in pairs]` with up to 2 x `workers` forwards in flight; yields the four stage maps of
        each pair in order (device tensors, complete when yielded)."""
        with self.pool(workers) as pool:
            window = []
            for left, right in pairs:
                window.append(pool.submit(left, right))
                if len(window) >= 2 * workers:
                    yield window.pop(0).result()
            for job in window:
                yield job.result()

    def _device_ctx(self):
        if self.device is None:
            import contextlib
            return contextlib.nullcontext()
        return torch.cuda.device(self.device)

    # ---- forward -------------------------------------------------------------------------------
    def _input(self, x, name):
        return as_input(x, name, self.device)

    def forward(self, left_input, right_input, out=None):
        """model(left, right) -> [pred1 .. pred4] (models.py:106-164).  `out` (not in the reference): optional list of
        four pre-allocated [B,1,H,W] device tensors (None entries are allocated) that receive the stage maps."""
        if self.device is None:
            raise RuntimeError("no HIP device is available and lwsnet_amd has no CPU fallback")
        if self._params is None:
            raise RuntimeError("set_state_dict() must be called before forward()")
        left = self._input(left_input, "left_input")
        right = self._input(right_input, "right_input")
        if left.shape != right.shape:
            raise ValueError(f"left/right shapes differ: {tuple(left.shape)} vs {tuple(right.shape)}")
        B, _, H, W = left.shape
        check_size(H, W, self.maxdisplist[0])
        with torch.cuda.device(self.device):
            return [DisparityTensor.wrap(p) for p in ops.forward(self._h, left, right, out)]      # models.py:106-164

    __call__ = forward


class _PoolJob:
    def __init__(self, pool, ticket, keep, outs):
        self._pool, self._ticket, self._keep, self._outs = pool, ticket, keep, outs

    def result(self):
        """Blocks until the four stage maps are complete in device memory and returns them."""
        self._pool._wait(self._ticket)
        self._keep = None
        return [DisparityTensor.wrap(o) for o in self._outs]


class ForwardPool:
    """Python face of lws_pool (see LWSNet.pool).  `submit` returns at once; `.result()` of the job waits for it."""

    def __init__(self, model, workers, side_streams=False):
        self._model = model                     # keeps the source handle alive
        self._lib = _lib.load()
        self._p = ctypes.c_void_p()
        self._shape = None
        self._live = {}                         # ticket -> (left, right, outs) of jobs that may still be running
        with torch.cuda.device(model.device):
            _lib.check(self._lib.lws_pool_create(model._h, workers, _lib.LWS_POOL_SIDE_STREAMS if side_streams else 0,
                                                 ctypes.byref(self._p)), "lws_pool_create")
        self.workers = workers

    def reserve(self, B, H, W):
        with torch.cuda.device(self._model.device):
            _lib.check(self._lib.lws_pool_reserve(self._p, int(B), int(H), int(W)), "lws_pool_reserve")
        self._shape = (B, H, W)
        return self

    def submit(self, left_input, right_input, out=None):
        m = self._model
        left, right = m._input(left_input, "left_input"), m._input(right_input, "right_input")
        if left.shape != right.shape:
            raise ValueError(f"left/right shapes differ: {tuple(left.shape)} vs {tuple(right.shape)}")
        B, _, H, W = left.shape
        check_size(H, W, m.maxdisplist[0])
        outs = ops.stage_outputs(out, B, H, W, left.device)          # validated: raw pointers go to a worker thread
        ticket = ctypes.c_int64(-1)
        with torch.cuda.device(m.device):
            if self._shape is None or B > self._shape[0] or (H, W) != tuple(self._shape[1:]):
                self.reserve(B, H, W)            # first use / new geometry: allocate before anything is in flight
            ptrs = (ctypes.c_void_p * 4)(*[o.data_ptr() for o in outs])
            after = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(self._lib.lws_pool_submit(self._p, ctypes.c_void_p(left.data_ptr()), ctypes.c_void_p(right.data_ptr()),
                                                 B, H, W, ptrs, after, ctypes.byref(ticket)), "lws_pool_submit")
        # the pool itself keeps the inputs and outputs alive until the ticket has completed: a job object dropped without
        # result() must not hand its tensors back to the caching allocator while a worker's stream still uses them.
        # lws_pool_submit recycles the slot of ticket t - 4 x workers only after hipEventSynchronize on that job, so every
        # ticket that old has left the device: a caller that never waits (fire-and-forget into `out`) does not grow this map
        self._live[ticket.value] = (left, right, outs)
        horizon = ticket.value - 4 * self.workers
        if horizon >= 0 and len(self._live) > 4 * self.workers:
            for t in [t for t in self._live if t <= horizon]:
                del self._live[t]
        return _PoolJob(self, ticket.value, (left, right), outs)

    def _wait(self, ticket):
        try:
            _lib.check(self._lib.lws_pool_wait(self._p, ctypes.c_int64(ticket)), "lws_pool_wait")
        finally:
            self._live.pop(ticket, None)

    def wait_all(self):
        try:
            _lib.check(self._lib.lws_pool_wait_all(self._p), "lws_pool_wait_all")
        finally:
            self._live.clear()

    def profile(self, class_mask, every_n=1):
        """Per-class kernel timing on every worker (lws_pool_profile_enable); call with nothing in flight."""
        _lib.check(self._lib.lws_pool_profile_enable(self._p, int(class_mask), int(every_n)), "lws_pool_profile_enable")

    def profile_read(self):
        """(total_ms[class], launches[class]) summed over the workers; call with nothing in flight."""
        tot = (ctypes.c_double * _lib.LWS_KC_COUNT)()
        cnt = (ctypes.c_int64 * _lib.LWS_KC_COUNT)()
        with torch.cuda.device(self._model.device):
            _lib.check(self._lib.lws_pool_profile_read(self._p, tot, cnt), "lws_pool_profile_read")
        return list(tot), list(cnt)

    def clear_error(self):
        """Clears the pool's sticky first-failure status (lws_pool_clear_error) so that submits are accepted again."""
        _lib.check(self._lib.lws_pool_clear_error(self._p), "lws_pool_clear_error")

    def close(self):
        if self._p:
            self._lib.lws_pool_destroy(self._p)          # runs what is queued, joins the workers, synchronises their streams
            self._p = ctypes.c_void_p()
        self._live.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
