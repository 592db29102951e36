#!/bin/bash
# round 6, call c: the pipelined CLI on host worker PROCESSES over shared-memory slots (call b's thread version saturated at
# ~120 pairs/s with 16 threads: the interpreter lock), its byte-identity test, and the end-to-end table again.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cli_directory or config1" > "$O/pytest_cli.txt" 2>&1; tail -12 "$O/pytest_cli.txt"
timeout 1500 python tools/e2e_cli.py --pairs 200 --workers 1 4 8 16 32 64 96 > "$O/e2e_cli.txt" 2> "$O/e2e_cli.err"; cat "$O/e2e_cli.txt"; tail -5 "$O/e2e_cli.err"
timeout 900 python tools/e2e_cli.py --pairs 400 --workers 64 128 --gpu_workers 4 > "$O/e2e_cli_gpu4.txt" 2>> "$O/e2e_cli.err"; cat "$O/e2e_cli_gpu4.txt"
df -h /dev/shm | tail -1
