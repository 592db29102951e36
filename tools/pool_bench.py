#!/usr/bin/env python3
"""lws_pool throughput at batch 1 for a matrix of worker counts x per-worker side streams (development aid).

    python tools/pool_bench.py [--size HxW] [--batch B] [--jobs N] [--workers 2,3,4,6] [--reps 3]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="256x512")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--jobs", type=int, default=900)
    ap.add_argument("--workers", default="2,3,4,6")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    H, W = [int(v) for v in a.size.split("x")]
    from lwsnet_amd.models import LWSNet
    from lwsnet_amd.synth import make_batch
    from lwsnet_amd.weights import default_args, make_state_dict
    dev = torch.device("cuda:0")
    m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
    for o in a.opt:
        m.set_option(o.split("=")[0], int(o.split("=")[1]))
    l, r = make_batch(a.batch, H, W, 0)
    l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    ref = [p.clone() for p in m(l, r)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        m(l, r)
    torch.cuda.synchronize()
    print(f"single stream: {a.batch * 200 / (time.perf_counter() - t0):8.1f} pairs/s")
    for side in (False, True):
        for P in [int(v) for v in a.workers.split(",")]:
            with m.pool(workers=P, side_streams=side) as pool:
                pool.reserve(a.batch, H, W)
                outs = [[torch.empty((a.batch, 1, H, W), device=dev) for _ in range(4)] for _ in range(2 * P)]
                rates = []
                for rep in range(a.reps + 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    jobs = []
                    for k in range(a.jobs):
                        if len(jobs) >= 2 * P:
                            jobs.pop(0).result()
                        jobs.append(pool.submit(l, r, out=outs[k % (2 * P)]))
                    last = [j.result() for j in jobs][-1]
                    dt = time.perf_counter() - t0
                    if rep:
                        rates.append(a.batch * a.jobs / dt)
                ok = all(bool(torch.equal(x, y)) for x, y in zip(last, ref))
                print(f"workers={P} side_streams={int(side)}: " + " ".join(f"{v:8.1f}" for v in rates) + f" pairs/s  bitwise={ok}", flush=True)


if __name__ == "__main__":
    main()
