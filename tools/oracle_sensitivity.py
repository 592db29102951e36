#!/usr/bin/env python3
"""How large is each bet the unpinned oracle makes on Paddle's op defaults?

The literal restatement (oracle/lws_oracle.py) keeps three readings of Paddle 2.0rc0 as switches -- `align_mode` of
F.interpolate, the grid_sample un-normalisation form (Paddle-CPU vs Paddle-CUDA), `tensor / scalar` as a multiply by the
reciprocal or a true division.  This tool runs the restatement in float64 (semantic difference, free of float32 noise) and in
float32 (what a float32 Paddle would show) under every single-switch variant and prints the per-stage max-abs / mean-abs
distance to the default reading, next to the float32 noise floor (default float32 vs default float64) for scale.  With
--reference it also runs the reference's OWN source behind tools/paddle_shim.py (which shares the switches) and checks that the
source under a variant equals the restatement under that variant.

    python tools/oracle_sensitivity.py [--sizes 64x256,256x512] [--reference /root/reference] > profiles/r04/oracle_sensitivity.txt
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from lwsnet_amd.synth import make_noise_pair, make_pair          # noqa: E402
from lwsnet_amd.weights import default_args, make_state_dict     # noqa: E402
from oracle import lws_oracle as O                               # noqa: E402

VARIANTS = [("align_mode=1 (Paddle 1.x / 2.0-beta interpolate: src = ratio*dst)", dict(align_mode=1)),
            ("grid_unnorm=cpu ((g+1)*((size-1)*0.5), Paddle's CPU grid_sample)", dict(grid_unnorm="cpu")),
            ("scalar_div=divide (tensor / scalar as a true division)", dict(scalar_div="divide"))]


def dist(a, b):
    return [(float((x.double() - y.double()).abs().max()), float((x.double() - y.double()).abs().mean())) for x, y in zip(a, b)]


def fmt(d):
    return "  ".join(f"s{i + 1} {m:9.3e} / {a:9.3e}" for i, (m, a) in enumerate(d))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="64x256,256x512")
    ap.add_argument("--reference", default=None, help="also run the reference's own source under each variant (build container only)")
    a = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    sd = make_state_dict(7)
    ml = default_args().maxdisplist
    print("per-stage distance to the DEFAULT reading (align_mode=0, grid_unnorm=cuda, scalar_div=reciprocal): max-abs / mean-abs, px")
    for size in a.sizes.split(","):
        H, W = [int(v) for v in size.split("x")]
        for kind in ("smooth", "noise"):
            if kind == "noise" and H * W > 64 * 256:
                continue
            left, right = (make_pair(H, W, 0)[:2] if kind == "smooth" else make_noise_pair(H, W, 0))
            l, r = left[None], right[None]
            base64 = O.forward(l, r, sd, ml, dtype=torch.float64)
            base32 = O.forward(l, r, sd, ml)
            print(f"\n== {H}x{W}, {kind} pair, seeded weights (calibrated BN); stage-4 disparity range "
                  f"{float(base64[3].min()):.1f} .. {float(base64[3].max()):.1f} px")
            print(f"  float32 noise floor (default fp32 vs default fp64)       : {fmt(dist(base32, base64))}")
            for name, kw in VARIANTS:
                with O.variant(**kw):
                    v64 = O.forward(l, r, sd, ml, dtype=torch.float64)
                    v32 = O.forward(l, r, sd, ml)
                print(f"  {name}")
                print(f"      float64 variant vs float64 default (semantic)          : {fmt(dist(v64, base64))}")
                print(f"      float32 variant vs float32 default (what fp32 shows)   : {fmt(dist(v32, base32))}")
                if a.reference and kind == "smooth" and H * W <= 64 * 256:
                    import check_oracle_vs_reference as chk
                    with O.variant(**kw):
                        ref32, _ = chk.run_reference(a.reference, default_args(), sd, l, r, torch.float32)
                    same = all(bool(torch.equal(torch.as_tensor(np.asarray(x)), y)) for x, y in zip(ref32, v32))
                    print(f"      reference source under this variant == restatement under this variant (fp32, bitwise): {same}")


if __name__ == "__main__":
    main()
