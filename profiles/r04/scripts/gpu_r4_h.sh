#!/bin/bash
# Round 4, call H: price a packed-float32 VALU form of the 8 -> 8 Conv3D layer (tools/micro/pkfma_conv.hip); re-run the tests
# the previous call stopped at.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4h
rm -rf "$O"; mkdir -p "$O"
cd "$R"
(cd tools/micro && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o pkfma_conv pkfma_conv.hip && ./pkfma_conv) > "$O/micro_pkfma_conv.txt" 2>&1; cat "$O/micro_pkfma_conv.txt"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
timeout 2400 python -m pytest tests -q -m gpu -k "split_bf16 or pool or dist or config1 or inference or profiler" > "$O/pytest_rest.txt" 2>&1; tail -6 "$O/pytest_rest.txt"
