"""Diagnostic: per-workgroup phase stamps of k_conv3d_mid16 (needs a library built with LWS_EXTRA_FLAGS=-DLWS_STAMPS).

    LWS_EXTRA_FLAGS=-DLWS_STAMPS python -m lwsnet_amd.build --force && python tools/stamps.py
Prints median cycles of: staging (0->1), MFMA loop (1->2), epilogue (2->3), and the spread of workgroup start/end times."""
import ctypes, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lwsnet_amd import _lib, ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device('cuda:0')
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
lib = ctypes.CDLL(_lib.LIB_PATH)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
c = torch.rand((B, 24, 32, 64), device=dev) * 12
for _ in range(5):
    ops.conv3d_stack(m._h, 0, c)
torch.cuda.synchronize()
n = 256 * 8
buf = (ctypes.c_ulonglong * n)()
assert lib.lws_debug_read_stamps(buf, n) == 0
s = np.array(buf, dtype=np.int64).reshape(-1, 8)[:, :4]
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
print("workgroups:", len(s))
print("start spread (cycles): p50 %d  max %d" % (np.median(s[:, 0] - t0), (s[:, 0] - t0).max()))
for a, b, name in ((0, 1, "staging"), (1, 2, "mfma loop"), (2, 3, "epilogue")):
    d = s[:, b] - s[:, a]
    print(f"{name:10s} median {np.median(d):8.0f}  p10 {np.percentile(d,10):8.0f}  p90 {np.percentile(d,90):8.0f} cycles (s_memtime = 100 MHz? see below)")
print("end spread: last end - first start = %d" % (s[:, 3].max() - t0))
