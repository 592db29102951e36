#!/bin/bash
# round 5, call o: the B=1 kernel trace again (the one inside the last full collection was host-bound under the profiler:
# 1,800 pairs/s, 4-12 us holes between the chain's kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5o
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt$i" -o run -- python3 "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined > "$O/bench_under_rocprof_$i.json" 2> "$O/kt$i.err"
  python3 "$R/tools/timeline.py" "$O/kt$i/run_kernel_trace.csv" 90 > "$O/timeline_b1_$i.txt" 2>&1
  python3 "$R/tools/overlap_account.py" "$O/kt$i/run_kernel_trace.csv" 90 8 > "$O/overlap_account_b1_256x512_$i.txt" 2>&1
  cp "$O/kt$i/run_kernel_stats.csv" "$O/kernel_stats_b1_256x512_$i.csv"; rm -rf "$O/kt$i"
  head -1 "$O/timeline_b1_$i.txt"; tail -3 "$O/overlap_account_b1_256x512_$i.txt"
  python3 -c "
import json; d=json.loads([l for l in open('$O/bench_under_rocprof_$i.json') if l.startswith('{')][-1]); print('under rocprof:', d['value'], d['ms_per_step'])"
done
