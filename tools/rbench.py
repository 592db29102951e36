#!/usr/bin/env python3
"""Per-kernel timing of the refinement (lws_refine) on the GPU, with per-launch k_ref_dws durations by dilation, refinement chunked (ref_chunk_mb = 72) and in one chunk (development aid).

    python tools/rbench.py [--batch B] [--size HxW] [--iters N]"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", default="256x512")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--opt", action="append", default=[], help="name=value launch-plan option (lws_set_option)")
    a = ap.parse_args()
    H, W = [int(v) for v in a.size.split("x")]
    from lwsnet_amd import _lib, ops
    from lwsnet_amd.models import LWSNet
    from lwsnet_amd.weights import default_args, make_state_dict
    dev = torch.device("cuda:0")
    m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
    lib = _lib.load()
    for o in a.opt:
        m.set_option(o.split("=")[0], int(o.split("=")[1]))
    left = torch.randn((a.batch, 3, H, W), device=dev)
    p3 = torch.rand((a.batch, 1, H, W), device=dev) * 100
    outs = {}
    for fuse in (72, 0):
        m.set_option("ref_chunk_mb", fuse)
        for _ in range(5):
            outs[fuse] = ops.refine(m._h, left, p3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            ops.refine(m._h, left, p3)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters
        _lib.check(lib.lws_profile_enable(m._h, -1))
        for _ in range(a.iters):
            ops.refine(m._h, left, p3)
        torch.cuda.synchronize()
        tot = (ctypes.c_double * _lib.LWS_KC_COUNT)()
        cnt = (ctypes.c_int64 * _lib.LWS_KC_COUNT)()
        _lib.check(lib.lws_profile_read(m._h, tot, cnt))
        per_launch = None
        each = (ctypes.c_float * 8192)()
        n_each = ctypes.c_int(0)
        _lib.check(lib.lws_profile_read_class(m._h, 10, each, 8192, ctypes.byref(n_each)))      # LWS_KC_REF_DWS
        per = n_each.value // a.iters
        if per:
            # launch order inside lws_refine: refinement1_left d = 2,4,8,16; refinement1_disp 2 (+ first conv),4,8,16; refinement2 8,4,2,1
            nch = max(1, per // 12)          # chunks: refinement1_left of every chunk first, then the rest of every chunk
            dil = [2, 4, 8, 16] * nch + [2, 4, 8, 16, 8, 4, 2, 1] * nch
            avg = [sum(each[k * per + j] for k in range(a.iters)) / a.iters * 1e3 for j in range(per)]
            per_launch = "   ref_dws per launch (dilation: us): " + "  ".join(f"d{d}:{u:.1f}" for d, u in zip(dil, avg))
        _lib.check(lib.lws_profile_enable(m._h, 0))
        print(f"ref_chunk_mb={fuse} B={a.batch} {H}x{W}: wall {wall * 1e6:.1f} us per lws_refine; kernels (with event overhead):")
        for kc in range(_lib.LWS_KC_COUNT):
            if cnt[kc]:
                print(f"   {lib.lws_kernel_class_name(kc).decode():12s} x{cnt[kc] // a.iters:2d} avg {tot[kc] / cnt[kc] * 1e3:7.2f} us")
        if per_launch:
            print(per_launch)
    print("bitwise equal:", bool(torch.equal(outs[72], outs[0])))


if __name__ == "__main__":
    main()
