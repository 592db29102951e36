#!/bin/bash
# round 6, call d: the fused first convolutions of refinement1 on fp32 MFMA (option "fuse_first" bit 2): parity of every form,
# then the A/B -- 1 = disparity branch fused on packed FMA (round 5's default), 5 = the same on MFMA, 3 / 7 = both branches fused
# on packed FMA / MFMA -- at B = 1, 8 and 8 x 368x1232, two passes; and what CPU budget this box gives the e2e pipeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6d
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "refine or schedule_options or profiler_counts or align_mode" > "$O/pytest_refine.txt" 2>&1; tail -8 "$O/pytest_refine.txt"
for pass in 1 2; do for ff in 1 5 3 7; do
  python bench.py --no-cpu-baseline --no-pipelined --steps 200 --warmup 20 --opt fuse_first=$ff > "$O/ab${pass}_ff${ff}_b1.json" 2> /dev/null
  python bench.py --batch 8 --no-cpu-baseline --no-pipelined --steps 40 --opt fuse_first=$ff > "$O/ab${pass}_ff${ff}_b8.json" 2> /dev/null
  python bench.py --batch 8 --size 368x1232 --no-cpu-baseline --no-pipelined --steps 12 --warmup 3 --opt fuse_first=$ff > "$O/ab${pass}_ff${ff}_cfg3.json" 2> /dev/null
done; done
python - <<PY > "$O/ab_fuse_first.txt"
import json, glob, os
print("# bench.py --opt fuse_first=N (bit 0: refinement1_disp's 1->32, bit 1: refinement1_left's 3->32 conv inside the first block; bit 2: on MFMA)")
print("# file value(pairs/s) ms/step mid16_us clock_ghz value/clock ref_first ref_dws(avg us over its launches)")
for f in sorted(glob.glob("$O/ab*_ff*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        k = d['kernels']; c = d['roofline'].get('clock_ghz') or 0
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], c, round(d['value'] / c, 1) if c else None,
              (k.get('ref_first') or {}).get('avg_us'), k['ref_dws'])
    except Exception as e:
        print(f, 'ERR', e)
PY
cat "$O/ab_fuse_first.txt"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"; cat /proc/loadavg
