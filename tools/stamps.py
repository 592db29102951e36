"""Diagnostic: per-workgroup phase stamps of the Conv3D MFMA kernels (library built with LWS_EXTRA_FLAGS=-DLWS_STAMPS).

    LWS_EXTRA_FLAGS=-DLWS_STAMPS python -m lwsnet_amd.build --force && python tools/stamps.py [stage] [batch]
Prints median shader cycles of: staging (0->1), MFMA loop (1->2), epilogue (2->3) of the LAST mid-layer launch."""
import ctypes, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lwsnet_amd import _lib, ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device('cuda:0')
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
lib = ctypes.CDLL(_lib.LIB_PATH)
what = sys.argv[1] if len(sys.argv) > 1 else "0"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if what == "feat":      # last k_conv2d_nchw launch of the feature extractor (classif1.2, 8 -> 8 at 1/2 resolution)
    TU = "conv2d"
    img = torch.randn((2 * B, 3, 256, 512), device=dev)
    for _ in range(3):
        ops.feature_extraction(m._h, img)
    stage = -1
elif what in ("dws", "conv64"):   # last k_ref_dws launch of the refinement (dilation 1) / the k_ref_conv64 launch
    TU = "conv2d"
    left = torch.randn((B, 3, 256, 512), device=dev)
    p3 = torch.rand((B, 1, 256, 512), device=dev) * 100
    for _ in range(3):
        ops.refine(m._h, left, p3)
    stage = -1
else:
    TU = "conv3d"
    stage = int(what)
    shape = [(B, 24, 32, 64), (B, 9, 64, 128), (B, 9, 128, 256)][stage]
    c = torch.rand(shape, device=dev) * 12
    for _ in range(5):
        ops.conv3d_stack(m._h, stage, c)
torch.cuda.synchronize()
n = 4096 * 8
buf = (ctypes.c_ulonglong * n)()
assert getattr(lib, "lws_debug_read_stamps_" + TU)(buf, n) == 0
NS = 4
raw = np.array(buf, dtype=np.int64).reshape(-1, 8)
s = raw[:, 4:8] if what == "conv64" else raw[:, :NS]
s = s[s[:, 0] > 0]
print(f"{what} B={B}: workgroups with stamps: {len(s)}")
phases = ((0, 1, "staging"), (1, 2, "depthwise"), (2, 3, "pointwise"), (0, 3, "total")) if what == "dws" else \
    ((0, 1, "staging"), (1, 2, "mfma loop"), (2, 3, "epilogue"), (0, 3, "total"))
for a, b, name in phases:
    d = s[:, b] - s[:, a]
    print(f"  {name:10s} median {np.median(d):8.0f}  p10 {np.percentile(d,10):8.0f}  p90 {np.percentile(d,90):8.0f} cycles")
