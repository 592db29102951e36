#!/usr/bin/env python3
"""Per-kernel timing of the hot path on the GPU (development aid, not the judged bench).

    python tools/kbench.py [--batch B] [--size HxW] [--iters N] [--check]

Prints the library profiler's average duration per kernel class for `lws_disparity_stages`
on seeded features, plus TFLOP/s for the MFMA kernels.  --check compares against the C oracle."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", default="256x512")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="name=value launch-plan option (lws_set_option)")
    a = ap.parse_args()
    H, W = [int(v) for v in a.size.split("x")]
    from lwsnet_amd import _lib, ops
    from lwsnet_amd.models import LWSNet
    from lwsnet_amd.weights import default_args, make_state_dict
    dev = torch.device("cuda:0")
    margs = default_args()
    sd = make_state_dict(7)
    m = LWSNet(margs, device=dev).set_state_dict(sd).eval()
    lib = _lib.load()
    for o in a.opt:
        k, v = o.split("=")
        m.set_option(k, int(v))
    B = a.batch
    rng = np.random.default_rng(0)
    H2, W2 = (H + 1) // 2, (W + 1) // 2          # the stem gives ceil(H/2); the hourglass halves twice more
    shapes = [(B, 16, H2 // 4, W2 // 4), (B, 16, H2 // 2, W2 // 2), (B, 8, H2, W2)]
    fl = [torch.from_numpy(np.abs(rng.standard_normal(s)).astype(np.float32) * 0.5).to(dev) for s in shapes]
    fr = [torch.from_numpy(np.abs(rng.standard_normal(s)).astype(np.float32) * 0.5).to(dev) for s in shapes]
    for _ in range(5):
        pred = ops.disparity_stages(m._h, fl, fr, H, W)
    torch.cuda.synchronize()
    _lib.check(lib.lws_profile_enable(m._h, -1))
    t0 = time.perf_counter()
    for _ in range(a.iters):
        pred = ops.disparity_stages(m._h, fl, fr, H, W)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters
    tot = (ctypes.c_double * _lib.LWS_KC_COUNT)()
    cnt = (ctypes.c_int64 * _lib.LWS_KC_COUNT)()
    _lib.check(lib.lws_profile_read(m._h, tot, cnt))
    _lib.check(lib.lws_profile_enable(m._h, 0))
    print(f"B={B} {H}x{W}: wall {wall * 1e6:.1f} us/iter (with event overhead), sum of kernels {sum(tot) / a.iters * 1e3:.1f} us")
    vox = [B * 24 * (H2 // 4) * (W2 // 4), B * 9 * (H2 // 2) * (W2 // 2), B * 9 * H2 * W2]
    for kc in range(_lib.LWS_KC_COUNT):
        if cnt[kc]:
            name = lib.lws_kernel_class_name(kc).decode()
            avg = tot[kc] / cnt[kc] * 1e3
            extra = ""
            if name == "conv3d_mid16":
                extra = f"  {2 * 27 * 32 * 32 * vox[0] / (avg * 1e-6) / 1e12:.1f} TF"
            if name == "conv3d_mid8":
                fl_ = 2 * 27 * 8 * 8 * (vox[1] + vox[2]) / 2
                extra = f"  {fl_ / (avg * 1e-6) / 1e12:.1f} TF (avg of stage 2 and 3)"
            print(f"  {name:16s} x{cnt[kc] // a.iters:2d}  avg {avg:8.2f} us{extra}")
    # without the profiler: pure wall per iteration
    t0 = time.perf_counter()
    for _ in range(a.iters):
        pred = ops.disparity_stages(m._h, fl, fr, H, W)
    torch.cuda.synchronize()
    print(f"  wall without profiler: {(time.perf_counter() - t0) / a.iters * 1e6:.1f} us/iter")
    if a.check:
        from oracle import c_oracle as C
        want = C.disparity_stages([f.cpu().numpy() for f in fl], [f.cpu().numpy() for f in fr], H, W, sd)
        for s in range(3):
            same = np.array_equal(pred[s].cpu().numpy(), want[s])
            print(f"  stage {s + 1} bit-exact vs C oracle: {same}")
            assert same


if __name__ == "__main__":
    main()
