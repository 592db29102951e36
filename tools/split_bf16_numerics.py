#!/usr/bin/env python3
"""Numerics account of option split_bf16 = 1 (k_conv3d_mid16x / k_conv3d_mid8x / k_ref_conv64x: split-bf16 MFMA in the Conv3D
middle layers and in refinement2[0]; --only mid16 | mid8 | conv64 for one of them (bits 1 / 2 / 4 of option split_bf16); k_conv3d_mid8x is only selected
for grids of >= 256 tiles, i.e. from about 128x384 up):
per stage, max and mean |result - float64 literal oracle| of (a) the exact HIP build (the oracle's float32 chain bit for bit),
(b) the split-bf16 build, (c) the float32 literal oracle, over several seeded pairs.  VERDICT r2 item 8: the split form is
acceptable only if it is no further from float64 than the float32 chain is.   python tools/split_bf16_numerics.py [--pairs N]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=6)
ap.add_argument("--size", default="64x256")
ap.add_argument("--only", default="", help="mid16, mid8 or conv64: switch only this kernel family (default: all three)")
a = ap.parse_args()
H, W = [int(v) for v in a.size.split("x")]
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_noise_pair, make_pair
from lwsnet_amd.weights import default_args, make_state_dict
from oracle import lws_oracle
dev = torch.device("cuda:0")
sd = make_state_dict(7)
m = LWSNet(default_args(), device=dev).set_state_dict(sd).eval()
torch.set_num_threads(16)


def set_mode(on):
    if a.only:
        m.set_option("split_bf16", {"mid16": 1, "mid8": 2, "conv64": 4}[a.only] if on else 0)
    else:
        m.set_option("split_bf16", 7 if on else 0)


acc = {k: {"max": np.zeros(4), "mean": np.zeros(4)} for k in ("exact HIP", "split-bf16 HIP", "float32 literal oracle")}
for i in range(a.pairs):
    if i == a.pairs - 1:
        l, r = make_noise_pair(H, W, 0)          # the adversarial white-noise pair last
    else:
        l, r, _ = make_pair(H, W, i)
    l, r = l[None], r[None]
    ref64 = lws_oracle.forward(l, r, sd, (24, 5, 5), dtype=torch.float64)
    ref32 = lws_oracle.forward(l, r, sd, (24, 5, 5))
    set_mode(False)
    exact = [p.cpu().double() for p in m(l, r)]
    set_mode(True)
    split = [p.cpu().double() for p in m(l, r)]
    set_mode(False)
    for name, res in (("exact HIP", exact), ("split-bf16 HIP", split), ("float32 literal oracle", [p.double() for p in ref32])):
        e = [(res[s] - ref64[s]).abs() for s in range(4)]
        mx, mn = np.array([float(v.max()) for v in e]), np.array([float(v.mean()) for v in e])
        acc[name]["max"] = np.maximum(acc[name]["max"], mx)
        acc[name]["mean"] += mn / a.pairs
        print(f"pair {i} {'(noise)' if i == a.pairs - 1 else '       '} {name:24s} max " + " ".join(f"{v:.3e}" for v in mx) + "   mean " + " ".join(f"{v:.3e}" for v in mn), flush=True)
print(f"\nover {a.pairs} pairs at {H}x{W} (px, stages 1-4):")
for name, v in acc.items():
    print(f"{name:24s} worst max " + " ".join(f"{x:.3e}" for x in v["max"]) + "   mean of means " + " ".join(f"{x:.3e}" for x in v["mean"]))
