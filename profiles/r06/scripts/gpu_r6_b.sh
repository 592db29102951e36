#!/bin/bash
# round 6, call b: the tests added after call a (pipelined CLI, profiler counts with the clock stamp), and the end-to-end
# directory throughput of the drop-in CLI on 200 copies of the reference pair.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6b
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cli_directory or profiler_counts or graph_capture or schedule_options" > "$O/pytest_new.txt" 2>&1; tail -12 "$O/pytest_new.txt"
timeout 1500 python tools/e2e_cli.py --pairs 200 --workers 1 2 4 8 16 32 64 > "$O/e2e_cli.txt" 2> "$O/e2e_cli.err"; cat "$O/e2e_cli.txt"; tail -5 "$O/e2e_cli.err"
timeout 600 python tools/e2e_cli.py --pairs 200 --workers 16 32 --gpu_workers 4 > "$O/e2e_cli_gpu4.txt" 2>> "$O/e2e_cli.err"; cat "$O/e2e_cli_gpu4.txt"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > "$O/bench_b1.json" 2> /dev/null; python -c "
import json
d=json.loads([l for l in open('$O/bench_b1.json') if l.startswith('{')][-1]); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['clock_ghz'])"
