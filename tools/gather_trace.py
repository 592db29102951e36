#!/usr/bin/env python3
"""Which kernels of the forward does RCCL's gather kernel delay?  Input: the rocprofv3 --kernel-trace CSV of
`tools/gather_probe.py --beside` (program directly after `--`, RANK / WORLD_SIZE / MASTER_* exported in the shell).
For every lws:: kernel: its median duration over launches that do NOT overlap an RCCL kernel against the median over the
launches that do; plus the RCCL kernels themselves (count, grid, duration).

    python tools/gather_trace.py <..._kernel_trace.csv>"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rccl = [r for r in rows if any(t in r["Kernel_Name"] for t in ("nccl", "rccl", "Rccl", "Nccl"))]
lws = [r for r in rows if "lws::" in r["Kernel_Name"]]
rccl.sort(key=lambda r: r["s"])
print(f"{len(rows)} kernel records: {len(lws)} lws::, {len(rccl)} RCCL")
by = collections.defaultdict(list)
for r in rccl:
    by[(r["Kernel_Name"].split("(")[0][:70], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))].append((r["e"] - r["s"]) / 1e3)
for (n, g, wg), v in by.items():
    print(f"  RCCL {n}  grid {g} wg {wg}: {len(v)} launches, median {statistics.median(v):.1f} us, max {max(v):.1f} us")
if not rccl:
    sys.exit(0)
# the last third of the RCCL launches are in the steady state of the last measured repetition
spans = [(r["s"], r["e"]) for r in rccl]
t_lo = spans[len(spans) // 2][0]


def overlap(r):
    return any(s < r["e"] and e > r["s"] for s, e in spans)


agg = collections.defaultdict(lambda: ([], []))
for r in lws:
    if r["s"] < t_lo:
        continue
    name = r["Kernel_Name"].replace("void lws::", "").replace("lws::", "").split("(")[0][:58] + " g" + r.get("Grid_Size", "?")
    agg[name][1 if overlap(r) else 0].append((r["e"] - r["s"]) / 1e3)
print(f"{'kernel':70s} {'alone: n':>9s} {'median us':>10s} | {'beside RCCL: n':>15s} {'median us':>10s} {'ratio':>6s}")
tot = 0.0
for name, (alone, with_) in sorted(agg.items(), key=lambda kv: -sum(kv[1][1])):
    if not with_ or not alone:
        continue
    ma, mw = statistics.median(alone), statistics.median(with_)
    tot += sum(with_) - ma * len(with_)
    print(f"{name:70s} {len(alone):9d} {ma:10.2f} | {len(with_):15d} {mw:10.2f} {mw / ma:6.2f}")
# the neighbourhood of one steady-state gather: every kernel of any queue from 150 us before the RCCL kernel to 250 us after
mid = rccl[(3 * len(rccl)) // 4]
print(f"\ntimeline around one RCCL kernel (t = 0 at its start; columns of the trace: {', '.join(k for k in rows[0].keys() if 'Id' in k)}):")
allk = sorted(rows, key=lambda r: r["s"])
last_end = {}
for r in allk:
    q = r.get("Queue_Id", "?")
    if mid["s"] - 150_000 <= r["s"] <= mid["e"] + 250_000:
        gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else 0.0
        name = r["Kernel_Name"].replace("void lws::", "").replace("lws::", "").split("(")[0][:48]
        print(f"  q{q:>3s} t={(r['s'] - mid['s']) / 1e3:8.1f} us dur {(r['e'] - r['s']) / 1e3:7.1f} gap_same_queue {gap:7.1f}  {name}")
    last_end[q] = r["e"]
n_g = sum(1 for s, e in spans if s >= t_lo)
print(f"extra kernel time beside RCCL kernels: {tot:.1f} us over {n_g} gathers = {tot / max(n_g, 1):.1f} us per gather")
