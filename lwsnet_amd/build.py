"""Builds lwsnet_amd/liblwsnet_hip.so (gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU.  -ffp-contract=off: the kernels and the host-side
BatchNorm folding must perform exactly the float32 operations written (the C oracle does
the same); fused multiply-adds appear only as explicit fmaf() / MFMA.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblwsnet_hip.so")
SOURCES = ["lws_api.hip", "lws_pool.hip", "lws_volume.hip", "lws_regress.hip", "lws_conv3d.hip", "lws_conv2d.hip", "lws_io.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "lwsnet_hip.h"),
                                                               os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, extra_flags=()):
    extra_flags = tuple(extra_flags) + tuple(os.environ.get("LWS_EXTRA_FLAGS", "").split())   # e.g. -DLWS_STAMPS (diagnostics)
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *extra_flags, *inc, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB + ".tmp"]
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
