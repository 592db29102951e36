#!/bin/bash
# round 6, call m: every development tool that drives the library still runs against ABI v8 (short invocations; output tails only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6m
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
t() { name=$1; shift; timeout 600 "$@" > "$O/$name.txt" 2>&1; echo "== $name rc=$?"; tail -4 "$O/$name.txt" | cut -c1-220; }
t kbench python tools/kbench.py
t fbench python tools/fbench.py
t hosttime python tools/hosttime.py 1
t graph_pipeline python tools/graph_pipeline.py
t thread_pipeline python tools/thread_pipeline.py
t gather_probe python tools/gather_probe.py
t split_numerics python tools/split_bf16_numerics.py --pairs 2
t split_numerics_only python tools/split_bf16_numerics.py --pairs 2 --only mid8
t sbench python tools/sbench.py
t rbench python tools/rbench.py --opt split_bf16=4
t pool_bench python tools/pool_bench.py --workers 2,4
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-pipelined --no-measure-traffic --opt split_bf16=7 > "$O/bench_b1_split_bf16.json" 2> "$O/split.err"; tail -2 "$O/split.err"
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --no-measure-traffic --opt split_bf16=7 > "$O/bench_b8_split_bf16.json" 2>> "$O/split.err"
for f in bench_b1_split_bf16 bench_b8_split_bf16; do python -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['dtype'][:50], d['roofline']['avg_launch_us'])"; done
