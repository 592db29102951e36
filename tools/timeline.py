#!/usr/bin/env python3
"""Condense a rocprofv3 kernel trace (…_kernel_trace.csv) of bench.py into the timeline of ONE steady-state forward:
kernel, stream (queue), start offset, duration, gap to the previous kernel on the same queue.  Usage:
    python tools/timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = [r for r in rows if "lws::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a forward starts with the first feature-head kernel k_conv2d_pair<3, 4, 8...
starts = [i for i, r in enumerate(rows) if "k_conv2d_pair<3, 4, 8" in r["Kernel_Name"]]
i0, i1 = starts[-back], starts[-back + 1]
t0 = int(rows[i0]["Start_Timestamp"])
last_end = {}
prev_end_any = t0
print(f"forward {len(starts) - back}: {i1 - i0} kernels, {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us to the next forward's first kernel")
for r in rows[i0:i1 + 3]:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void lws::", "").replace("lws::", "").split("(")[0]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    print(f"q{q:>3s} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap_same_queue {gap:7.1f}  {name[:60]}")
    last_end[q] = e
