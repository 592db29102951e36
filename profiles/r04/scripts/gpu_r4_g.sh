#!/bin/bash
# Round 4, call G: the complete GPU test suite, smoke, then the profile run profiles/r04 is built from.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4g
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > "$O/pytest_gpu.txt" 2>&1; tail -6 "$O/pytest_gpu.txt"
bash tools/profile_run.sh > "$O/profile_run.log" 2>&1; tail -30 "$O/profile_run.log"
