#!/bin/bash
# round 6, call k: stability of the round's final library -- the soak (thousands of back-to-back forwards into rotating output
# buffers, bit-compared with the first result), the pipelined CLI on 1,000 pairs, the CLI test incl. the forced staging path, and
# the driver-flags line with frac_at_held_clock.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6k
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cli_directory" > "$O/pytest_cli.txt" 2>&1; tail -3 "$O/pytest_cli.txt"
timeout 900 python tools/soak.py > "$O/soak.txt" 2>&1; tail -6 "$O/soak.txt"
timeout 900 python tools/e2e_cli.py --pairs 1000 --workers 12 > "$O/e2e_cli_1000.txt" 2> "$O/e2e.err"; cat "$O/e2e_cli_1000.txt"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_b1_driver_flags.json" 2> /dev/null; python -c "
import json
d=json.loads([l for l in open('$O/bench_b1_driver_flags.json') if l.startswith('{')][-1]); r=d['roofline']
print(d['value'], r['frac'], r['clock_ghz'], r['frac_at_held_clock'], r['traffic'], r['traffic_measured_in_run'])"
