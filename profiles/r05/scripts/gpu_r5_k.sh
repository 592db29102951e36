#!/bin/bash
# round 5, call k: overlap accounts again with the corrected estimator (slowdown booked to the overlapped part), host time
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5k
rm -rf "$O"; mkdir -p "$O"
cd /tmp; export TMPDIR=/tmp
for cfg in "b1_256x512:--steps 60 --warmup 10:90:8" "b8_256x512:--batch 8 --steps 20 --warmup 5:70:4" "b8_368x1232:--batch 8 --size 368x1232 --steps 8 --warmup 3:64:4"; do
  tag=${cfg%%:*}; rest=${cfg#*:}; args=${rest%%:*}; rest=${rest#*:}; back=${rest%%:*}; cnt=${rest##*:}
  rocprofv3 --kernel-trace --output-format csv -d "$O/kt_$tag" -o run -- python3 "$R/bench.py" $args --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  python3 "$R/tools/timeline.py" "$O/kt_$tag/run_kernel_trace.csv" $back > "$O/timeline_${tag}.txt" 2>&1
  python3 "$R/tools/overlap_account.py" "$O/kt_$tag/run_kernel_trace.csv" $back $cnt > "$O/overlap_account_${tag}.txt" 2>&1
  rm -rf "$O/kt_$tag"
done
cd "$R"
python tools/hosttime.py 1 > "$O/hosttime_b1.txt" 2>&1
for t in b1_256x512 b8_256x512 b8_368x1232; do tail -4 "$O/overlap_account_$t.txt"; done; tail -6 "$O/hosttime_b1.txt"
