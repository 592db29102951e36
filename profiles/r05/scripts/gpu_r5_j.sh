#!/bin/bash
# round 5, call j: DEBUG_CLR_SKIP_RELEASE_SCOPE=1 -- cross-XCD visibility micro-test with and without it, then the whole GPU
# parity suite under it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5j
rm -rf "$O"; mkdir -p "$O"
cd "$R"
hipcc --offload-arch=gfx950 -O3 -o /tmp/release_scope tools/micro/release_scope.hip
timeout 300 /tmp/release_scope > "$O/micro_release_scope.txt" 2>&1
DEBUG_CLR_SKIP_RELEASE_SCOPE=1 timeout 300 /tmp/release_scope >> "$O/micro_release_scope.txt" 2>&1
cat "$O/micro_release_scope.txt"
DEBUG_CLR_SKIP_RELEASE_SCOPE=1 timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$O/pytest_parity_skip_release.txt" 2>&1; tail -5 "$O/pytest_parity_skip_release.txt"
