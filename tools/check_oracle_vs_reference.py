#!/usr/bin/env python3
"""Run the reference's OWN source (/root/reference/models/models.py, imported in place) on torch-CPU through a
stand-in for the ~25 Paddle calls it makes (tools/paddle_shim.py), and

  1. compare all four stage maps with the hand restatement oracle/lws_oracle.py (float32 and float64), for the default
     constructor arguments and two non-default ones -- this executes the reference's control flow, layer wiring,
     padding rules and state-dict names from the reference's text, so it removes the transcription risk of the
     230-line restatement;
  2. with --write, store the stage maps as tests/golden/ref_source_*.npz (inputs + float32 outputs + float64 outputs).
     Provenance of those vectors: reference source text + torch-CPU kernels + the Paddle op defaults listed in
     tools/paddle_shim.py (from memory of Paddle 2.0).  They are NOT PaddlePaddle outputs: the
     oracle remains "parity unpinned" (DESIGN.md section 2).

  3. with --check-fixtures NAME..., re-run the reference's source and compare with the COMMITTED fixtures bit for bit
     (tests/test_reference_source_cpu.py starts this mode as a child process: the reference's files are public,
     untrusted content, so they are never imported into the pytest process).

  4. with --real-paddle (VERDICT r4 item 5 -- the only route from "parity unpinned" to a pinned oracle): when `import paddle`
     finds a REAL PaddlePaddle, the stand-in is NOT installed; the reference's source runs on Paddle-CPU on the five committed
     cases and the report lists, per stage, |paddle - committed ref_source float32| beside the float32 noise floor of the
     committed vectors (|ref_source float32 - ref_source float64|), the gate being the one the HIP build is held to
     (|paddle - fp64| <= 1.25 x floor + 1e-4 px); then one Paddle-written `.pdparams` (`paddle.save(model.state_dict())`) is
     read back through lwsnet_amd.checkpoint.load_state_dict and compared array by array.  Per case it also says WHICH reading
     of F.interpolate the Paddle at hand follows: its stage maps against the literal restatement in float64 under
     align_mode = 0 (half-pixel centres, this build's default) and align_mode = 1 (src = ratio * dst), each against that
     reading's own float32 noise floor.  If every case sits inside the gate under align_mode = 1 and not under 0, the exit
     status is 3 and the report says what to do: build the model with `interp_align_mode = 1` (lws_config / the args
     namespace) -- a configuration flip, not a kernel change.  No Paddle wheel exists in this
     container, so this mode has only ever run against a fake `paddle` package in tests/test_host_cpu.py; the one command for
     a machine that has Paddle 2.0:  python -B tools/check_oracle_vs_reference.py --real-paddle --reference <LWSNet checkout>

Build-container only: /root/reference does not exist on the GPU box and nothing here is imported by the product,
bench.py or smoke(); the tests only ever start it as a subprocess.  Nothing from /root/reference is copied (the fixtures
are numeric arrays) and nothing is written there (no bytecode: sys.dont_write_bytecode).

Usage: python -B tools/check_oracle_vs_reference.py [--write | --check-fixtures NAME... | --real-paddle] [--reference /root/reference]
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True                       # /root/reference is read-only content: leave no __pycache__ in it

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import paddle_shim                                   # noqa: E402
from lwsnet_amd.synth import make_noise_pair, make_pair   # noqa: E402
from lwsnet_amd.weights import default_args, make_state_dict   # noqa: E402
from oracle import lws_oracle as O                   # noqa: E402

CASES = [
    # name, H, W, pair kind, constructor arguments, calibrated BN
    ("e2e_64x256", 64, 256, "smooth", dict(), True),
    ("e2e_noise_64x256", 64, 256, "noise", dict(), True),
    ("e2e_d32_64x320", 64, 320, "smooth", dict(maxdisplist=(32, 5, 5)), True),
    ("e2e_args_32x256", 32, 256, "smooth", dict(maxdisplist=(24, 3, 4), layers_3d=3, channels_3d=8, growth_rate=(2, 1, 1)), False),
    ("e2e_odd_63x255", 63, 255, "smooth", dict(), True),      # H, W = 8k-1: legal for the reference (ceil(H/2) % 4 == 0)
]
# ... and one case under the OTHER reading of F.interpolate (align_mode = 1, src = ratio * dst): the stand-in and the
# restatement share the switch (oracle.lws_oracle.VARIANT), the product has it as lws_config.interp_align_mode
VARIANT_CASES = [("e2e_align1_64x256", 64, 256, "smooth", dict(), True, dict(align_mode=1))]


def case_variant(name):
    for c in VARIANT_CASES:
        if c[0] == name:
            return c[6]
    return {}


def run_reference(ref_root, args, sd, left, right, dtype):
    """LWSNet from the reference's source, float32 or float64 (the hard-coded dtype='float32' strings map to `dtype`)."""
    paddle_shim.set_dtype(dtype)
    paddle_shim.install()
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[m]
    from models.models import LWSNet                 # the reference's file, imported in place
    src = sys.modules["models.models"].__file__
    assert os.path.realpath(src).startswith(os.path.realpath(ref_root)), src
    model = LWSNet(args)
    model.set_state_dict(sd)                         # raises on any key / shape mismatch with the reference's own layers
    model.eval()
    with torch.no_grad():
        out = model(paddle_shim.to_tensor(left, "float32"), paddle_shim.to_tensor(right, "float32"))
    assert isinstance(out, list) and len(out) == 4
    return [o.numpy() for o in out], sorted(model.state_dict().keys())


def real_paddle():
    """The real PaddlePaddle module if `import paddle` finds one (never the stand-in), else None."""
    if "paddle" in sys.modules and getattr(sys.modules["paddle"], "_lws_shim", False):
        raise RuntimeError("the stand-in is already installed in this process: --real-paddle must run in a fresh one")
    try:
        import paddle
    except ImportError:
        return None
    return None if getattr(paddle, "_lws_shim", False) else paddle


def run_reference_on_paddle(paddle, ref_root, args, sd, left, right):
    """LWSNet from the reference's source on the real Paddle (CPU place), float32 as published."""
    paddle.set_device("cpu")
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    for m in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[m]
    from models.models import LWSNet                 # the reference's file, imported in place
    assert not getattr(sys.modules["paddle"], "_lws_shim", False), "the stand-in must not be installed in --real-paddle mode"
    model = LWSNet(args)
    model.set_state_dict({k: v for k, v in sd.items()})
    model.eval()
    with paddle.no_grad():
        out = model(paddle.to_tensor(left, dtype="float32"), paddle.to_tensor(right, dtype="float32"))
    assert isinstance(out, list) and len(out) == 4
    return [np.asarray(o.numpy(), dtype=np.float32) for o in out], model


def check_real_paddle(ref_root, keep_pdparams=None):
    """--real-paddle: exit status 0 = every case inside the noise-floor gate and the .pdparams round trip exact;
    1 = a case outside the gate under both readings of F.interpolate (parity broken or another Paddle default read wrongly --
    see tools/oracle_sensitivity.py); 2 = no real PaddlePaddle importable (nothing was checked); 3 = every case inside the
    gate under align_mode = 1 only: this Paddle resizes with src = ratio * dst -- set interp_align_mode = 1."""
    paddle = real_paddle()
    if paddle is None:
        print("--real-paddle: `import paddle` failed (or found the stand-in): no PaddlePaddle here, nothing checked; "
              "the oracle stays 'parity unpinned'")
        return 2
    print(f"--real-paddle: PaddlePaddle {getattr(paddle, '__version__', '?')} from {getattr(paddle, '__file__', '?')}; stand-in NOT installed")
    bad = 0
    model = sd = None
    inside = {0: 0, 1: 0}                                # cases inside the gate under each reading of F.interpolate's align_mode
    for name, H, W, kind, kw, calib in CASES:
        with np.load(os.path.join(ROOT, "tests", "golden", f"ref_source_{name}.npz")) as z:
            g = {k: z[k] for k in z.files}
        args = default_args(**kw)
        sd = make_state_dict(int(g["seed"]), args, calibrated=bool(g["calibrated"]))
        out, model = run_reference_on_paddle(paddle, ref_root, args, sd, g["left"], g["right"])
        for mode in (0, 1):
            with O.variant(align_mode=mode):
                lit64 = [p.numpy() for p in O.forward(g["left"], g["right"], sd, args.maxdisplist, torch.float64)]
                lit32 = [p.numpy() for p in O.forward(g["left"], g["right"], sd, args.maxdisplist, torch.float32)]
            dm = [float(np.abs(out[i].astype(np.float64) - lit64[i]).max()) for i in range(4)]
            fm = [float(np.abs(lit32[i].astype(np.float64) - lit64[i]).max()) for i in range(4)]
            okm = all(dm[i] <= 1.25 * fm[i] + 1e-4 for i in range(4))
            inside[mode] += 1 if okm else 0
            print(f"{name:18s} align_mode {mode}: |paddle - literal fp64| {['%.3e' % v for v in dm]}  floor {['%.3e' % v for v in fm]}"
                  f"  -> {'INSIDE' if okm else 'outside'}")
        d32 = [float(np.abs(out[i] - g[f"pred{i}"]).max()) for i in range(4)]
        floor = [float(np.abs(g[f"pred{i}"].astype(np.float64) - g[f"pred64_{i}"]).max()) for i in range(4)]
        d64 = [float(np.abs(out[i].astype(np.float64) - g[f"pred64_{i}"]).max()) for i in range(4)]
        ok = all(d64[i] <= 1.25 * floor[i] + 1e-4 for i in range(4))
        bits = all(np.array_equal(out[i], g[f"pred{i}"]) for i in range(4))
        print(f"{name:18s} |paddle - committed ref_source fp32| per stage {['%.3e' % v for v in d32]}{' (bit-equal)' if bits else ''}")
        print(f"{'':18s} |paddle - fp64| {['%.3e' % v for v in d64]}  noise floor |ref_source fp32 - fp64| {['%.3e' % v for v in floor]}"
              f"  -> {'INSIDE' if ok else 'OUTSIDE'} the gate (<= 1.25 x floor + 1e-4 px)")
        bad += 0 if ok else 1
    # one Paddle-written .pdparams through the product's loader (inference.py:45 reads what train.py:115 writes)
    import tempfile
    from lwsnet_amd import checkpoint
    with tempfile.TemporaryDirectory() as td:
        path = keep_pdparams or os.path.join(td, "real_paddle.pdparams")
        paddle.save(model.state_dict(), path)
        got = checkpoint.load_state_dict(path)
        same = sorted(got) == sorted(sd) and all(np.array_equal(got[k], np.asarray(sd[k], dtype=np.float32)) for k in sd)
        print(f".pdparams written by paddle.save ({os.path.getsize(path)} bytes) read by lwsnet_amd.checkpoint.load_state_dict: "
              f"{len(got)} entries, {'all equal to what was set' if same else 'DIFFERS'}")
        bad += 0 if same else 1
    n = len(CASES)
    if inside[0] == n:
        print("F.interpolate of this PaddlePaddle follows align_mode = 0 (half-pixel centres) on every case: interp_align_mode = 0, "
              "the default of lws_config and of every committed fixture, is the right setting")
    elif inside[1] == n:
        print("F.interpolate of this PaddlePaddle follows align_mode = 1 (src = ratio * dst) on every case, NOT the default this build "
              "bets on.  What to do: construct the model with interp_align_mode = 1 (lws_config.interp_align_mode; "
              "`args.interp_align_mode = 1` for lwsnet_amd.models.LWSNet) -- a configuration flip, the kernels and the C oracle "
              "carry both readings (tests/test_gpu_parity.py::test_interp_align_mode_bitexact_vs_c_oracle); the committed "
              "ref_source_* fixtures then describe the other reading and the checks above report OUTSIDE for that reason only")
        return 3
    else:
        print(f"F.interpolate: inside the gate on {inside[0]} / {n} cases under align_mode = 0 and {inside[1]} / {n} under align_mode = 1: "
              "neither reading explains this Paddle (tools/oracle_sensitivity.py lists the other switches)")
    print("--real-paddle:", "parity with PaddlePaddle holds on the committed cases" if not bad else f"{bad} check(s) FAILED")
    return 1 if bad else 0


def check_fixtures(ref_root, names):
    """The committed tests/golden/ref_source_<name>.npz must be what the reference's source produces now, bit for bit."""
    bad = 0
    for name in names:
        with np.load(os.path.join(ROOT, "tests", "golden", f"ref_source_{name}.npz")) as z:
            g = {k: z[k] for k in z.files}
        args = default_args(maxdisplist=tuple(int(v) for v in g["maxdisplist"]), layers_3d=int(g["layers_3d"]),
                            channels_3d=int(g["channels_3d"]), growth_rate=tuple(int(v) for v in g["growth_rate"]))
        sd = make_state_dict(int(g["seed"]), args, calibrated=bool(g["calibrated"]))
        with O.variant(**case_variant(name)):            # (a fixture made under another reading of Paddle's defaults says so)
            out, keys = run_reference(ref_root, args, sd, g["left"], g["right"], torch.float32)
        ok = keys == sorted(sd.keys()) and all(np.array_equal(out[i], g[f"pred{i}"]) for i in range(4))
        print(f"{name}: {'OK' if ok else 'DIFFERS'} (226 structured names {'match' if keys == sorted(sd.keys()) else 'DIFFER'})")
        bad += 0 if ok else 1
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--check-fixtures", nargs="+", metavar="NAME")
    ap.add_argument("--real-paddle", action="store_true",
                    help="run the reference's source on a REAL PaddlePaddle (no stand-in) against the committed fixtures; "
                         "exit 2 when `import paddle` fails")
    ap.add_argument("--keep-pdparams", metavar="PATH", help="--real-paddle: keep the Paddle-written checkpoint here")
    a = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    if a.real_paddle:
        return check_real_paddle(a.reference, a.keep_pdparams)
    if a.check_fixtures:
        return 1 if check_fixtures(a.reference, a.check_fixtures) else 0
    worst = 0.0
    for name, H, W, kind, kw, calib, *var in CASES + VARIANT_CASES:
        var = var[0] if var else {}
        args = default_args(**kw)
        sd = make_state_dict(7, args, calibrated=calib)
        if kind == "noise":
            l, r = make_noise_pair(H, W, 0)
        else:
            l, r, _ = make_pair(H, W, 0)
        l, r = l[None], r[None]
        with O.variant(**var):
            ref32, keys = run_reference(a.reference, args, sd, l, r, torch.float32)
            ref64, _ = run_reference(a.reference, args, sd, l, r, torch.float64)
            assert keys == sorted(sd.keys()), "state-dict names differ from the reference's own layers"
            ora32 = [p.numpy() for p in O.forward(l, r, sd, args.maxdisplist, torch.float32)]
            ora64 = [p.numpy() for p in O.forward(l, r, sd, args.maxdisplist, torch.float64)]
        d32 = [float(np.abs(x - y).max()) for x, y in zip(ref32, ora32)]
        d64 = [float(np.abs(x - y).max()) for x, y in zip(ref64, ora64)]
        n32 = [float(np.abs(x.astype(np.float64) - y).max()) for x, y in zip(ref32, ref64)]
        print(f"{name:18s} reference-source vs restatement, max-abs per stage: fp32 {d32}  fp64 {d64}")
        print(f"{'':18s} reference-source fp32 vs its own fp64 run (noise floor): {n32}")
        worst = max(worst, max(d32), max(d64))
        if a.write:
            out = os.path.join(ROOT, "tests", "golden", f"ref_source_{name}.npz")
            np.savez_compressed(out, left=l, right=r, seed=7, calibrated=calib, maxdisplist=np.array(args.maxdisplist),
                                layers_3d=args.layers_3d, channels_3d=args.channels_3d, growth_rate=np.array(args.growth_rate),
                                align_mode=int(var.get("align_mode", 0)),
                                **{f"pred{i}": p.astype(np.float32) for i, p in enumerate(ref32)},
                                **{f"pred64_{i}": p.astype(np.float64) for i, p in enumerate(ref64)})
            print("  wrote", out, os.path.getsize(out))
    print("worst difference reference-source vs restatement:", worst)
    return 0 if worst == 0.0 else 1


if __name__ == "__main__":
    sys.exit(main())
