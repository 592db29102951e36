#!/usr/bin/env python3
"""Is the side-stream overlap inside a forward worth anything?  

Reads a `rocprofv3 --kernel-trace` CSV of bench.py (trace only: kernels of different queues DO overlap in it) and, for the
forwards of the timed region, accounts every kernel launch as

    in-situ duration            end - start as traced
    alone duration              the median duration of the launches of the SAME kernel with the SAME grid that overlapped no
                                kernel of another queue (falls back to the smallest in-situ duration when there is none)
    overlapped                  whether a kernel of another queue ran during it (and for how long)

and then sums, over the windows in which two queues were busy at once,

    wall        the length of those windows
    work        the alone-equivalent progress the kernels made inside them: for every kernel, the time it spent inside such
                windows minus its excess over its alone duration (a kernel runs at its alone rate while nothing runs beside it,
                so all of its slowdown is booked to the overlapped part)

work / wall = 1.0 means the overlap is ZERO-SUM (the two kernels time-share the chip: running them back to back would take as
long); 2.0 would mean both ran at their alone rate.  Per kernel class it prints launches, in-situ and alone microseconds per
forward and the slowdown.  Usage:

    python tools/overlap_account.py <kernel_trace.csv> [first_forward_from_end] [forwards]
"""
import csv
import statistics
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void lws::", "").replace("lws::", "")
    return n.split("(")[0][:48]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 70
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rows = [r for r in csv.DictReader(open(path)) if "lws::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = []
    for r in rows:
        grid = tuple(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Grid_Size"))
        ks.append({"name": short(r["Kernel_Name"]), "key": (short(r["Kernel_Name"]), grid), "q": r.get("Queue_Id", "?"),
                   "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])})
    for i, k in enumerate(ks):
        k["i"] = i
    starts = [i for i, k in enumerate(ks) if k["name"].startswith("k_conv2d_pair<3, 4, 8")]
    if len(starts) < back + 1:
        raise SystemExit(f"only {len(starts)} forwards in the trace")
    i0, i1 = starts[-back], starts[-back + count] if back > count else len(ks)
    sel = ks[i0:i1]
    # overlap of every selected kernel with kernels of other queues (also just outside the selection)
    ctx = ks[max(0, i0 - 80):min(len(ks), i1 + 80)]
    for k in sel:
        ov = 0
        for o in ctx:
            if o["q"] != k["q"] and o["e"] > k["s"] and o["s"] < k["e"]:
                ov += min(o["e"], k["e"]) - max(o["s"], k["s"])
        k["ov"] = ov
        k["dur"] = k["e"] - k["s"]
    # alone durations from the whole trace (more samples than the selection)
    pool = defaultdict(list)
    for k in ks:
        pool[k["key"]].append(k)
    alone = {}
    for key, lst in pool.items():
        free = []
        for k in lst:
            hit = False
            # (neighbours in start order: a linear scan around the kernel is enough for a test like this)
            for o in ks[max(0, k["i"] - 12):k["i"] + 12]:
                if o is not k and o["q"] != k["q"] and o["e"] > k["s"] and o["s"] < k["e"]:
                    hit = True
                    break
            if not hit:
                free.append(k["e"] - k["s"])
        alone[key] = (statistics.median(free), len(free)) if free else (min(k["e"] - k["s"] for k in lst), 0)
    per = defaultdict(lambda: [0, 0.0, 0.0, 0])
    for k in sel:
        a = alone[k["key"]][0]
        p = per[k["name"]]
        p[0] += 1
        p[1] += k["dur"] / 1e3
        p[2] += a / 1e3
        p[3] += 1 if k["ov"] > 0 else 0
    nf = count
    print(f"{path}: forwards {len(starts) - back} .. {len(starts) - back + count - 1} of {len(starts)} ({len(sel)} kernels)")
    print(f"{'kernel':50s} {'launches':>8s} {'overlapped':>10s} {'in situ us':>11s} {'alone us':>9s} {'slowdown':>8s}   (per forward)")
    for name, (n, d, a, o) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:50s} {n / nf:8.1f} {o / nf:10.1f} {d / nf:11.1f} {a / nf:9.1f} {d / a if a else 0:8.2f}")
    tot_d = sum(v[1] for v in per.values()) / nf
    tot_a = sum(v[2] for v in per.values()) / nf
    # union busy time and two-queue windows
    ev = []
    for k in sel:
        ev.append((k["s"], 1, k))
        ev.append((k["e"], -1, k))
    ev.sort(key=lambda t: (t[0], t[1]))
    active = []
    last = ev[0][0]
    busy = two = 0
    for k in sel:
        k["in2"] = 0                       # time this kernel spent inside two-queue windows
    for t, d, k in ev:
        if active:
            busy += t - last
            if len({a["q"] for a in active}) >= 2:
                two += t - last
                for a in active:
                    a["in2"] += t - last
        last = t
        if d == 1:
            active.append(k)
        else:
            active.remove(k)
    work_two = 0.0
    for k in sel:
        excess = max(0.0, k["dur"] - alone[k["key"]][0])
        work_two += max(0.0, k["in2"] - excess)
    span = (sel[-1]["e"] - sel[0]["s"]) / 1e3 / nf
    print(f"per forward: span {span:.1f} us, some queue busy {busy / 1e3 / nf:.1f} us, two queues busy {two / 1e3 / nf:.1f} us")
    print(f"             sum of in-situ durations {tot_d:.1f} us, sum of alone durations {tot_a:.1f} us")
    if two:
        print(f"inside the two-queue windows: alone-equivalent work {work_two / 1e3 / nf:.1f} us in {two / 1e3 / nf:.1f} us of wall time "
              f"= {work_two / two:.2f}x  (1.0 = zero-sum time sharing, 2.0 = both at their alone rate)")


if __name__ == "__main__":
    main()
