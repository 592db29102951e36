#!/bin/bash
# Round 4, call J: a kernel trace of the B=1 bench WITHOUT --stats (the --stats run serialises the queues), timeline of one forward.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4j
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -o run -- python3 "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined > "$O/bench_under_trace.json" 2> "$O/kt.err"
python3 "$R/tools/timeline.py" $(find "$O/kt" -name '*kernel_trace.csv' | head -1) 90 > "$O/timeline_b1_trace_only.txt" 2>&1
python3 "$R/tools/timeline.py" $(find "$O/kt" -name '*kernel_trace.csv' | head -1) 85 > "$O/timeline_b1_trace_only_2.txt" 2>&1
rm -rf "$O/kt"
cat "$O/timeline_b1_trace_only.txt"; tail -c 300 "$O/bench_under_trace.json"
