#!/usr/bin/env python3
"""Conv3D stack of one stage alone (lws_conv3d_stack): per-kernel-class averages, k_conv3d_mid8q (4x4x1_16B, exact) vs k_conv3d_mid8x (option split_bf16 = 2, not bit-exact) (dev aid)."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lwsnet_amd import _lib, ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.weights import default_args, make_state_dict
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--size", default="256x512")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
H, W = [int(v) for v in a.size.split("x")]
dev = torch.device("cuda:0")
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
lib = _lib.load()
opts = ' '.join(a.opt)
for o in a.opt:
    m.set_option(o.split('=')[0], int(o.split('=')[1]))
for stage, (D, div) in ((1, (9, 4)), (2, (9, 2))):
    c = torch.rand((a.batch, D, H // div, W // div), device=dev) * 12
    for form in (0, 2):                      # option "split_bf16": 0 = k_conv3d_mid8q (exact), 2 = k_conv3d_mid8x (not bit-exact)
        m.set_option("split_bf16", form)
        for _ in range(5):
            ops.conv3d_stack(m._h, stage, c)
        torch.cuda.synchronize()
        _lib.check(lib.lws_profile_enable(m._h, -1))
        for _ in range(a.iters):
            ops.conv3d_stack(m._h, stage, c)
        torch.cuda.synchronize()
        tot = (ctypes.c_double * _lib.LWS_KC_COUNT)()
        cnt = (ctypes.c_int64 * _lib.LWS_KC_COUNT)()
        _lib.check(lib.lws_profile_read(m._h, tot, cnt))
        _lib.check(lib.lws_profile_enable(m._h, 0))
        kc = 4
        avg = tot[kc] / cnt[kc] * 1e3
        gf = 2 * 27 * 8 * 8 * a.batch * D * (H // div) * (W // div)
        print(f"stage {stage + 1} B={a.batch} {opts} split_bf16={form}: k_conv3d_mid8{'x' if form else 'q'} avg {avg:7.2f} us = {gf / avg / 1e6:6.1f} TF useful")
