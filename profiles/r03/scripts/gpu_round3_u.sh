set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3u; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -o run -- python3 "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined > "$O/kt_bench.json" 2> "$O/kt.err"
cd $R
python tools/timeline.py $(ls $O/kt/*/run_kernel_trace.csv $O/kt/run_kernel_trace.csv 2>/dev/null | head -1) 10 > $O/timeline_b1.txt 2>&1
python tools/timeline.py $(ls $O/kt/*/run_kernel_trace.csv $O/kt/run_kernel_trace.csv 2>/dev/null | head -1) 20 > $O/timeline_b1_2.txt 2>&1
rm -rf $O/kt
cat $O/timeline_b1.txt
