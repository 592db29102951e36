#!/bin/bash
# round 6, call l: the end-to-end table of the CLI at its final form (worker start-up outside the clock, the minimal PNG writer),
# 200 and 1,000 pairs; the CLI tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6l
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cli_directory or config1 or io_kernels" > "$O/pytest_cli.txt" 2>&1; tail -3 "$O/pytest_cli.txt"
timeout 1500 python tools/e2e_cli.py --pairs 200 --workers 1 4 8 12 16 > "$O/e2e_cli.txt" 2> "$O/e2e.err"; cat "$O/e2e_cli.txt"
timeout 1500 python tools/e2e_cli.py --pairs 1000 --workers 12 14 > "$O/e2e_cli_1000.txt" 2>> "$O/e2e.err"; cat "$O/e2e_cli_1000.txt"
