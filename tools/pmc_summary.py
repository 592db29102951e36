#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc passes (counter_collection CSVs) into the per-kernel JSON summaries committed under profiles/rNN.
Runs on the GPU box right after the passes (the raw CSVs are tens of MB; gpurun merges back at most 64 MiB).

    python tools/pmc_summary.py <out_dir> <tag> <workload description> <fetch_dir> <write_dir> [<sq_dir>]

writes <out_dir>/pmc_fetch_write_<tag>.json and, with <sq_dir>, <out_dir>/pmc_sq_<tag>.json.  FETCH_SIZE and WRITE_SIZE come
from SEPARATE passes (MI355X_MICROARCH.md, HBM section); values are KB per launch; on gfx950 FETCH_SIZE under-reports wide
(16 B/lane) streaming reads by 2x, so corrected HBM-side bytes = 2*FETCH + WRITE (applied by the reader, bench.py)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_csv(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    return f[0]


def kname(r):
    return r["Kernel_Name"].split("(")[0].replace("void ", "") + " grid=" + r["Grid_Size"]


def main():
    out_dir, tag, what, fetch_dir, write_dir = sys.argv[1:6]
    sq_dir = sys.argv[6] if len(sys.argv) > 6 else None
    os.makedirs(out_dir, exist_ok=True)
    sha = {n: hashlib.sha256(open(os.path.join(ROOT, "lwsnet_amd", "csrc", n), "rb").read()).hexdigest()
           for n in sorted(os.listdir(os.path.join(ROOT, "lwsnet_amd", "csrc")))}
    out = {}
    for name, d in (("FETCH", fetch_dir), ("WRITE", write_dir)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(counter_csv(d))):
            if "lws::" in r["Kernel_Name"] and r["Counter_Name"] == name + "_SIZE":
                agg[kname(r)].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            e = out.setdefault(k, {})
            e[name + "_SIZE_KB_avg"] = round(sum(v) / len(v), 2)
            e["launches"] = len(v)
    for k, e in out.items():
        if "FETCH_SIZE_KB_avg" in e and "WRITE_SIZE_KB_avg" in e:
            e["hbm_bytes_corrected"] = round((2.0 * e["FETCH_SIZE_KB_avg"] + e["WRITE_SIZE_KB_avg"]) * 1024.0)
    json.dump({"workload": what, "kernel_source_sha256": sha,
               "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains), python3 bench.py. Raw "
                       "counter values in KB per launch; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) streaming reads by "
                       "2x (MI355X_MICROARCH.md, HBM section): hbm_bytes_corrected = (2*FETCH + WRITE) * 1024.",
               "kernels": out}, open(os.path.join(out_dir, f"pmc_fetch_write_{tag}.json"), "w"), indent=1)
    if sq_dir:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(counter_csv(sq_dir))):
            if "lws::" in r["Kernel_Name"]:
                k = kname(r)
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "End_Timestamp" in r and r.get("End_Timestamp"):
                    agg[k]["duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        sq = {k: {c: round(sum(x) / len(x), 1) for c, x in v.items()} for k, v in sorted(agg.items())}
        for k, v in sq.items():
            v["launches"] = len(agg[k]["SQ_WAVE_CYCLES"]) if "SQ_WAVE_CYCLES" in agg[k] else None
        json.dump({"workload": what, "kernel_source_sha256": sha,
                   "note": "rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE, averages per launch (duration_ns is the profiled launch: counters "
                           "serialise kernels). SQ_VALU_MFMA_BUSY_CYCLES: summed over the chip's 1024 SIMDs, 32 per v_mfma_f32_16x16x4_f32; "
                           "SQ_INSTS_VALU_MFMA_F32 counts MFMA instructions issued per wave; SQ_BUSY_CYCLES per shader engine.",
                   "kernels": sq}, open(os.path.join(out_dir, f"pmc_sq_{tag}.json"), "w"), indent=1)
    print("wrote", out_dir, tag, len(out), "kernels")


if __name__ == "__main__":
    main()
