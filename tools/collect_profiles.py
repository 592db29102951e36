#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of a profile run (gpurun_out/final) into the committed summaries under profiles/rNN."""
import collections, csv, json, os, shutil, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03"
os.makedirs(dst, exist_ok=True)
for f in os.listdir(dst):
    if f != "README.md" and os.path.isfile(os.path.join(dst, f)):      # (experiments/ holds outputs of other runs: kept)
        os.remove(os.path.join(dst, f))
shutil.copy(f"{src}/kt/run_kernel_stats.csv", f"{dst}/kernel_stats_b1_256x512.csv")
shutil.copy(f"{src}/kt_bench.json", f"{dst}/bench_under_rocprof.json")
for n in os.listdir(src):
    if (n.startswith("bench_") and n.endswith(".json")) or n.startswith(("sbench_", "rbench_", "pool_bench_", "micro_", "stamps_", "split_bf16_", "gather_probe")):
        shutil.copy(f"{src}/{n}", f"{dst}/{n}")
for d, name in (("kt8", "kernel_stats_b8_256x512.csv"), ("ktc3", "kernel_stats_b8_368x1232.csv")):
    if os.path.isfile(f"{src}/{d}/run_kernel_stats.csv"):
        shutil.copy(f"{src}/{d}/run_kernel_stats.csv", f"{dst}/{name}")
out = {}
for name in ("fetch", "write"):
    rows = list(csv.DictReader(open(f"{src}/{name}/run_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        if "lws::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (k, g), v in sorted(agg.items()):
        e = out.setdefault(f"{k} grid={g}", {})
        e[name.upper() + "_SIZE_KB_avg"] = round(sum(v) / len(v), 2)
        e["launches"] = len(v)
import hashlib
sha = {n: hashlib.sha256(open(f"lwsnet_amd/csrc/{n}", "rb").read()).hexdigest() for n in ("lws_conv3d.hip", "lws_conv2d.hip")}
json.dump({"kernel_source_sha256": sha, "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), `python3 bench.py --steps 5 --warmup 2 "
           "--no-cpu-baseline`, B=1 256x512. Raw counter values in KB per launch; on gfx950 FETCH_SIZE under-reports wide "
           "(16 B/lane) streaming reads by 2x (MI355X_MICROARCH.md, HBM section): corrected HBM-side bytes = 2*FETCH + WRITE.",
           "kernels": out}, open(f"{dst}/pmc_fetch_write_b1_256x512.json", "w"), indent=1)
rows = list(csv.DictReader(open(f"{src}/sq/run_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "lws::" in r["Kernel_Name"]:
        key = r["Kernel_Name"].split("(")[0].replace("void ", "") + " grid=" + r["Grid_Size"]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
sq = {k: {c: round(sum(x) / len(x), 1) for c, x in v.items()} for k, v in agg.items() if any(t in k for t in ("mid", "conv64", "dws"))}
json.dump({"note": "rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE, B=1 256x512, averages per launch. SQ_VALU_MFMA_BUSY_CYCLES = 32 per "
           "v_mfma_f32_16x16x4_f32 (sum over 1024 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles.",
           "kernels": sq}, open(f"{dst}/pmc_sq_b1_256x512.json", "w"), indent=1)
for n in sorted(os.listdir(dst)):
    if n.startswith("bench_") and n.endswith(".json"):
        d = json.loads(open(f"{dst}/{n}").read().strip().splitlines()[-1])
        cb = d.get("cpu_baseline") or {}
        print(f"{n:28s} {d['value']:8.1f} pairs/s {d['ms_per_step']:8.3f} ms  mid16 {d['roofline']['achieved']:6.1f} TF "
              f"({d['roofline']['avg_launch_us']} us)  cpu {cb.get('value')}")
rows = list(csv.DictReader(open(f"{dst}/kernel_stats_b1_256x512.csv")))
for r in rows[:6]:
    print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
