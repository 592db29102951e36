#!/usr/bin/env python3
"""Feature extractor alone (lws_feature_extraction) on N images: wall time per call (development aid; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lwsnet_amd import ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.weights import default_args, make_state_dict
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--size", default="256x512")
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
H, W = [int(v) for v in a.size.split("x")]
dev = torch.device("cuda:0")
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
img = torch.randn((a.n, 3, H, W), device=dev)
for _ in range(5):
    ops.feature_extraction(m._h, img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    ops.feature_extraction(m._h, img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
gf = 0.243e9 * a.n * (H * W) / (256 * 512)
print(f"feature_extraction N={a.n} {H}x{W}: {dt * 1e6:.1f} us per call = {gf / dt / 1e12:.2f} TFLOP/s")
