#!/usr/bin/env python3
"""End-to-end directory throughput of the drop-in CLI (/root/reference/inference.py:50-63,78-138).

    python tools/e2e_cli.py --pairs 200 --workers 1 4 8 16 32 > profiles/r06/e2e_cli.txt

Builds a KITTI-style folder (image_2/ + image_3/) from the reference's own pair (tests/golden/kitti_pair/, BASELINE config 1)
-- every copy shifted by a few columns so that no two files hold the same image -- and runs the loop of
`python -m lwsnet_amd.inference --img_path DIR` over it: once sequentially (the reference's loop: one pair at a time on one
thread, which is what its published "10 FPS" measures) and once per --workers value (host worker processes) through the pipelined path.  Prints pairs/s
end to end (decode -> PNG on disk) and where the time goes; checks that every pipelined run wrote byte-identical files.
Seeded synthetic weights (the reference ships no checkpoint)."""
import argparse
import hashlib
import logging
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lwsnet_amd  # noqa: E402,F401  (runtime switches before HIP is up)
import numpy as np  # noqa: E402


def make_folder(root, pairs):
    from PIL import Image
    kp = os.path.join(ROOT, "tests", "golden", "kitti_pair")
    l0 = np.asarray(Image.open(os.path.join(kp, "left_test.png")).convert("RGB"))
    r0 = np.asarray(Image.open(os.path.join(kp, "right_test.png")).convert("RGB"))
    for d in ("image_2", "image_3"):
        os.makedirs(os.path.join(root, d))
    for i in range(pairs):
        Image.fromarray(np.roll(l0, 3 * i, axis=1)).save(os.path.join(root, "image_2", f"{i:06d}_10.png"))
        Image.fromarray(np.roll(r0, 3 * i, axis=1)).save(os.path.join(root, "image_3", f"{i:06d}_10.png"))


def digest(folder):
    h = hashlib.sha256()
    names = sorted(os.listdir(folder))
    for n in names:
        with open(os.path.join(folder, n), "rb") as f:
            h.update(n.encode() + f.read())
    return len(names), h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=200)
    ap.add_argument("--workers", type=int, nargs="+", default=[1, 4, 8, 16, 32])
    ap.add_argument("--gpu_workers", type=int, default=3)
    a = ap.parse_args()
    import glob
    import torch
    from lwsnet_amd import inference as inf
    from lwsnet_amd.models import LWSNet
    from lwsnet_amd.weights import make_state_dict
    logging.basicConfig(stream=sys.stderr, level=logging.WARNING)
    log = logging.getLogger("e2e")
    tmp = tempfile.mkdtemp(prefix="lws_e2e_")
    try:
        make_folder(tmp, a.pairs)
        lefts = sorted(glob.glob(os.path.join(tmp, "image_2/*.png")))
        rights = sorted(glob.glob(os.path.join(tmp, "image_3/*.png")))
        args = inf.build_parser().parse_args(["--img_path", tmp, "--save_path", os.path.join(tmp, "out_seq"), "--synthetic_weights"])
        dev = torch.device("cuda", 0)
        model = LWSNet(args, device=dev).set_state_dict(make_state_dict(7, args)).eval()
        quota = "?"
        for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            try:
                quota = f"{f}: {open(f).read().strip()}"
                break
            except OSError:
                pass
        print(f"# {a.pairs} pairs 1242x375 -> crop 368x1232, maxdisplist {args.maxdisplist}, host has {os.cpu_count()} logical CPUs "
              f"(affinity {len(os.sched_getaffinity(0))}, cgroup quota {quota}), torch {torch.__version__}; seeded synthetic weights")
        os.makedirs(args.save_path)
        t0 = time.perf_counter()
        inf.inference(model, lefts, rights, args, log)
        torch.cuda.synchronize()
        seq = time.perf_counter() - t0
        nseq, dseq = digest(args.save_path)
        print(f"sequential loop (inference.py:88-137, one thread): {nseq} files, {seq:.2f} s = {a.pairs / seq:7.1f} pairs/s "
              f"({1e3 * seq / a.pairs:.1f} ms per pair)")
        # isolated forward, as the sequential loop times it
        l1 = torch.zeros((1, 3, 368, 1232), device=dev)
        ts = []
        for _ in range(30):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model(l1, l1)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t1))
        print(f"isolated forward 1 x 368x1232 (host call -> stream idle): median {sorted(ts)[15]:.3f} ms")
        print("workers gpu_workers pairs/s  vs_seq  decode_ms encode_ms h2d_ms d2h_ms latency_ms shm_pinned files_identical")
        for w in a.workers:
            args.workers, args.gpu_workers = w, a.gpu_workers
            args.save_path = os.path.join(tmp, f"out_w{w}")
            os.makedirs(args.save_path)
            _, st = inf.inference_pipelined(model, lefts, rights, args, log)
            n, d = digest(args.save_path)
            print(f"{w:7d} {a.gpu_workers:11d} {st['pairs_per_s']:7.1f} {st['pairs_per_s'] * seq / a.pairs:6.2f}x {st['decode_ms_per_pair']:9.2f} "
                  f"{st['encode_ms_per_pair']:9.2f} {st['h2d_ms_per_pair']:6.3f} {st['d2h_ms_per_pair']:6.3f} {st['latency_ms_per_pair']:10.1f} {str(st['shared_memory_pinned']):>10s} "
                  f"{n == nseq and d == dseq}")
            shutil.rmtree(args.save_path)
        print("# workers = host worker processes; decode / encode: their time per pair (PNG decode + crop of two images; PNG encode of one); "
              "h2d: upload of 2 x 1.36 MB uint8 + lws_preprocess_rgb8, d2h: lws_apply_lut8 + download of 1.36 MB (hipEvent-timed); "
              "latency: decode start -> file on disk")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
