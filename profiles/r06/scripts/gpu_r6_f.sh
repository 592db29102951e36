#!/bin/bash
# round 6, call f: what the driver runs at round end, at HEAD (build, GPU suite, smoke, default bench with the r06 PMC summary in
# place so that roofline.traffic resolves), plus the split-bf16 lines the profile run left empty.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6f
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 3000 python -m pytest tests -x -q -m gpu > "$O/pytest.txt" 2>&1; tail -6 "$O/pytest.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_b1_driver_flags.json" 2> "$O/bench_b1.err"; python -c "
import json
d=json.loads([l for l in open('$O/bench_b1_driver_flags.json') if l.startswith('{')][-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['clock_ghz'], r['traffic'], r.get('traffic_source'), d['cpu_baseline']['value'])"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt split_bf16=7 > "$O/bench_b1_split_bf16.json" 2> "$O/split.err"; tail -5 "$O/split.err"
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt split_bf16=7 > "$O/bench_b8_split_bf16.json" 2>> "$O/split.err"
for f in bench_b1_split_bf16 bench_b8_split_bf16; do python -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['dtype'][:40], d['roofline']['avg_launch_us'])"; done
