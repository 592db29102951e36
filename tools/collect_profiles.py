#!/usr/bin/env python3
"""Copies the summaries a profile run left under gpurun_out/final (tools/profile_run.sh) into profiles/rNN and prints the bench
lines.  It only ever writes the files it copies: whatever else lives in the destination (experiments, notes, evidence of other
runs) stays -- round 3's version deleted every file it had not produced, which is how cited evidence went missing.

    python tools/collect_profiles.py [gpurun_out/final] [profiles/r05]"""
import csv
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05"
os.makedirs(dst, exist_ok=True)
PREFIXES = ("bench_", "sbench_", "rbench_", "wbench", "pool_bench_", "micro_", "stamps_", "kernel_stats_", "pmc_", "timeline_", "gather_", "overlap_account_")
copied = []
for n in sorted(os.listdir(src)):
    p = os.path.join(src, n)
    if os.path.isfile(p) and n.startswith(PREFIXES) and not n.endswith(".err") and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, n))
        copied.append(n)
print(f"copied {len(copied)} files into {dst}")
for n in copied:
    if n.startswith("bench_") and n.endswith(".json"):
        try:
            d = json.loads(open(os.path.join(dst, n)).read().strip().splitlines()[-1])
        except Exception as e:
            print(f"{n:36s} unreadable: {e}")
            continue
        cb = d.get("cpu_baseline") or {}
        r = d.get("roofline") or {}
        if d.get("value") is None:                       # --one-gpu / --dry-run-cpu lines carry no throughput
            print(f"{n:36s} value null ({(d.get('shared_one_gpu') or {}).get('pairs_per_s_both_ranks_on_one_gpu')} pairs/s with the ranks sharing one GPU)")
            continue
        print(f"{n:36s} {d['value']:8.1f} pairs/s {d['ms_per_step']:8.3f} ms  mid16 {r.get('achieved')} TF ({r.get('avg_launch_us')} us) "
              f"frac {r.get('frac')} step_frac {r.get('step_frac')} traffic {r.get('traffic')}  cpu {cb.get('value')}")
f = os.path.join(dst, "kernel_stats_b1_256x512.csv")
if os.path.isfile(f):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
