#!/usr/bin/env python3
"""Per-launch timing of k_volume_l1_warp (lws_volume_l1_warp through the C ABI) at the stage-2 / stage-3 shapes of a
workload, on a smooth synthetic disparity (4 ... 70 px + noise, the shape of bench.py's pairs) -- development aid.

    python tools/wbench.py [--batch B] [--size HxW] [--iters N] [--noise PX]

Prints, per stage: algorithmic bytes (read L, R, the previous map once, write the volume once: SURVEY.md section 8d), the
average launch-to-launch time of N back-to-back launches (events on the stream) and the GB/s that is.  Run it under
`rocprofv3 --kernel-trace --stats` for kernel-only durations."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", default="256x512")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--noise", type=float, default=1.0, help="uniform noise on the previous disparity, full-resolution px")
    a = ap.parse_args()
    H, W = [int(v) for v in a.size.split("x")]
    from lwsnet_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B = a.batch
    rng = np.random.default_rng(0)
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    yy, xx = np.mgrid[0:H, 0:W]
    prev = (4 + 60 * yy / H + 6 * np.sin(2 * np.pi * xx / W * 3))[None, None].repeat(B, 0) + rng.random((B, 1, H, W)) * a.noise
    prev = torch.from_numpy(prev.astype(np.float32)).to(dev)
    for name, C, h, w in (("stage 2", 16, H2 // 2, W2 // 2), ("stage 3", 8, H2, W2)):
        L = torch.from_numpy(rng.standard_normal((B, C, h, w)).astype(np.float32)).to(dev)
        R = torch.from_numpy(rng.standard_normal((B, C, h, w)).astype(np.float32)).to(dev)
        nbytes = (2 * C * h * w + 9 * h * w + H * W) * 4 * B          # L, R, previous full-resolution map, 9-plane volume
        nbytes_survey = (2 * C * h * w + 9 * h * w + h * w) * 4 * B    # SURVEY 8d counts wflow at low resolution
        # the C ABI directly on pre-allocated outputs: the Python wrapper's allocation + checks cost ~12 us per call, more than
        # the kernel at small sizes
        cost = torch.empty((B, 9, h, w), device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def launch():
            _lib.check(lib.lws_volume_l1_warp(L.data_ptr(), R.data_ptr(), prev.data_ptr(), cost.data_ptr(), None, B, C, h, w, H, W, 5, st),
                       "lws_volume_l1_warp")

        for _ in range(10):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"B={B} {H}x{W} {name}: [{B},{C},{h},{w}] algorithmic {nbytes_survey / 1e6:.2f} MB (SURVEY 8d; {nbytes / 1e6:.2f} MB with the "
              f"full-resolution map): {us:.2f} us per launch back to back = {nbytes_survey / us / 1e3:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
