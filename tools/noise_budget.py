"""Where is the float32 noise of the stage maps born?  (CPU experiment, literal oracle.)

Runs the literal restatement (oracle/lws_oracle.py) in float64 with ONE component at a time in float32 (inputs cast
down, result cast up) and reports each stage's max-abs distance to the all-float64 run.  The all-float32 line is the
reference algorithm's own noise floor.  Usage: python tools/noise_budget.py [--size 256x512] [--noise]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwsnet_amd.synth import make_noise_pair, make_pair  # noqa: E402
from lwsnet_amd.weights import make_state_dict  # noqa: E402
from oracle import lws_oracle as O  # noqa: E402

COMPONENTS = ["features", "volume1", "conv3d_1", "regress1", "volume2", "conv3d_2", "regress2", "volume3", "conv3d_3",
              "regress3", "refine"]


def forward_mixed(left, right, sd, low=()):
    """float64 everywhere except the components named in `low` (float32)."""
    f64, f32 = torch.float64, torch.float32

    def dt(name):
        return f32 if name in low else f64

    left64 = torch.as_tensor(left, dtype=f64)
    right64 = torch.as_tensor(right, dtype=f64)
    H, W = left64.shape[2:]
    d = dt("features")
    fl = [f.to(f64) for f in O.feature_extraction(left64.to(d), sd, d)]
    fr = [f.to(f64) for f in O.feature_extraction(right64.to(d), sd, d)]
    pred = []
    mdl = [24, 5, 5]
    for s in range(3):
        h, w = fl[s].shape[2:]
        d = dt(f"volume{s + 1}")
        if s == 0:
            cost = O.build_volume_2d(fl[0].to(d), fr[0].to(d), mdl[0], d).to(f64)
        else:
            wflow = O._scale(O._interp(pred[s - 1].to(d), [h, w]) * float(h), H, d)
            cost = O.build_volume_2d3(fl[s].to(d), fr[s].to(d), mdl[s], wflow, d).to(f64)
        d = dt(f"conv3d_{s + 1}")
        c5 = cost.to(d).unsqueeze(1)
        cost = (O.post_3dconvs(c5, sd, s, d) + c5).squeeze(1).to(f64)
        d = dt(f"regress{s + 1}")
        p = F.softmax(-cost.to(d), dim=1)
        lowd = O.disparity_regression(p, 0 if s == 0 else -mdl[s] + 1, mdl[s], d)
        lowd = O._scale(lowd * float(H), lowd.shape[2], d)
        up = O._interp(lowd, [H, W])
        up = up if s == 0 else up + pred[s - 1].to(d)
        pred.append(up.to(f64))
    d = dt("refine")
    pred.append(O.refine(left64.to(d), pred[2].to(d), sd, d).to(f64))
    return pred


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="256x512")
    ap.add_argument("--noise", action="store_true")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    H, W = (int(v) for v in a.size.split("x"))
    if a.noise:
        l, r = make_noise_pair(H, W, 0)
    else:
        l, r, _ = make_pair(H, W, 0)
    sd = make_state_dict(7)
    with torch.no_grad():
        ref = forward_mixed(l[None], r[None], sd, ())

        def report(name, low):
            p = forward_mixed(l[None], r[None], sd, low)
            e = [float((a_ - b_).abs().max()) for a_, b_ in zip(p, ref)]
            print(f"{name:12s} " + " ".join(f"{v:9.2e}" for v in e), flush=True)

        print(f"{'fp32 part':12s} " + " ".join(f"stage{s + 1:>4d}" for s in range(4)))
        report("ALL", tuple(COMPONENTS))
        for c in COMPONENTS:
            report(c, (c,))


if __name__ == "__main__":
    main()
