#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the literal oracle (oracle/lws_oracle.py, float32).

PARITY UNPINNED: these vectors come from this repository's own restatement of the
reference, not from PaddlePaddle (not installable here).  They pin
the restatement against regressions and give the GPU tests fixed inputs.  Inputs are
stored next to the expected outputs, so nothing is regenerated on the GPU box.
Run from the repo root:  python tools/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lwsnet_amd.synth import make_pair            # noqa: E402
from lwsnet_amd.weights import make_state_dict    # noqa: E402
from oracle import lws_oracle as O                # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(2024)
    sd = make_state_dict(7)

    # K1: stage-1 volume, [1,16,8,32], D = 24 (incl. occlusion columns x < d)
    L = rng.standard_normal((1, 16, 8, 32)).astype(np.float32)
    R = rng.standard_normal((1, 16, 8, 32)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "volume_shift.npz"), L=L, R=R, D=24,
                        cost=O.build_volume_2d(t(L), t(R), 24).numpy())

    # K2: residual volume, [1,16,16,64], m = 5, prev disparity with out-of-range samples
    L = rng.standard_normal((1, 16, 16, 64)).astype(np.float32)
    R = rng.standard_normal((1, 16, 16, 64)).astype(np.float32)
    H, W = 64, 256
    # +100 px at the left border (samples fall left of the image) down to -40 px at the right border
    # (samples fall right of it), plus noise so that fractional positions are generic
    ramp = 100.0 - 140.0 * np.arange(W, dtype=np.float64)[None, None, None, :] / W
    prev = (ramp + rng.random((1, 1, H, W)) * 8.0).astype(np.float32)
    wflow = O._scale(O._interp(t(prev), [16, 64]) * 16.0, H, torch.float32)
    cost = O.build_volume_2d3(t(L), t(R), 5, wflow)
    np.savez_compressed(os.path.join(OUT, "volume_warp.npz"), L=L, R=R, prev=prev, m=5,
                        wflow=wflow.numpy()[:, 0], cost=cost.numpy())

    # K3: conv3d stacks with the seeded weights (stage 1: c3 = 32 on [1,24,8,32]; stage 2: c3 = 8 on [1,9,8,16])
    for stage, shape in ((0, (1, 24, 8, 32)), (1, (1, 9, 8, 16))):
        c = (rng.random(shape) * 12.0).astype(np.float32)
        y = O.post_3dconvs(t(c)[:, None], sd, stage)[:, 0] + t(c)
        np.savez_compressed(os.path.join(OUT, f"conv3d_stage{stage}.npz"), cost_in=c, cost_out=y.numpy(), seed=7)

    # K4/K5: soft-argmin (D = 24, start 0; D = 9, start -4) and upsample+add
    for name, D, start in (("softargmin_d24", 24, 0), ("softargmin_d9", 9, -4)):
        c = (rng.standard_normal((1, D, 8, 16)) * 4.0).astype(np.float32)
        p = torch.softmax(-t(c), 1)
        low = O.disparity_regression(p, start, start + D)
        H, W = 64, 128
        prev = (rng.random((1, 1, H, W)) * 50).astype(np.float32)
        up = O._interp(O._scale(low * float(H), 8, torch.float32), [H, W]) + t(prev)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), cost=c, start=start, low=low.numpy()[:, 0],
                            prev=prev, up=up.numpy())

    # end to end, tiny 1x3x64x256 (W/8 = 32 > 24)
    l, r, _ = make_pair(64, 256, 0)
    with torch.no_grad():
        fl = O.feature_extraction(t(l[None]), sd)
        fr = O.feature_extraction(t(r[None]), sd)
    pred = O.forward(l[None], r[None], sd)
    np.savez_compressed(os.path.join(OUT, "e2e_64x256.npz"), left=l[None], right=r[None], seed=7,
                        **{f"featL{i}": f.numpy() for i, f in enumerate(fl)},
                        **{f"featR{i}": f.numpy() for i, f in enumerate(fr)},
                        **{f"pred{i}": p.numpy() for i, p in enumerate(pred)})
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
