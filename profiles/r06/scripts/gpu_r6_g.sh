#!/bin/bash
# round 6, call g: tile shapes of k_conv3d_last at large batches (experiment switches LWS_LAST_TILE / LWS_LAST8_TILE, removed
# after the A/B): the C3 = 32 layer on 6x4x16 / 3x8x16 / 3x4x32 / 2x4x16 tiles instead of 3x4x16, the fused C3 = 8 layer on
# 9x4x16 / 9x2x32 / 9x3x32 instead of 9x2x16 -- fewer halo bytes per output against fewer workgroups per CU.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6g
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
run() { # tag, env assignment, bench args
  env $2 python bench.py --no-cpu-baseline --no-pipelined $3 > "$O/$1.json" 2> "$O/$1.err"
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); k=d['kernels']
    print('$1', d['value'], d['ms_per_step'], 'last', k['conv3d_last'], 'clk', d['roofline']['clock_ghz'], 'v/clk', round(d['value']/d['roofline']['clock_ghz'],1))
except Exception as e: print('$1 ERR', e, open('$O/$1.err').read()[-300:])"
}
for pass in 1 2; do
  for t in 0 1 2 3 4; do run "p${pass}_b8_last32_$t" "LWS_LAST_TILE=$t" "--batch 8 --steps 40"; done
  for t in 0 1 2 3; do run "p${pass}_b8_last8_$t" "LWS_LAST8_TILE=$t" "--batch 8 --steps 40"; done
  for t in 0 1 2 3 4; do run "p${pass}_cfg3_last32_$t" "LWS_LAST_TILE=$t" "--batch 8 --size 368x1232 --steps 12 --warmup 3"; done
  for t in 0 1 2 3; do run "p${pass}_cfg3_last8_$t" "LWS_LAST8_TILE=$t" "--batch 8 --size 368x1232 --steps 12 --warmup 3"; done
done 2>&1 | tee "$O/ab_last_tiles.txt"
LWS_LAST_TILE=1 LWS_LAST8_TILE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3d_stack or batch_paths or full_size" 2>&1 | tail -3
LWS_LAST_TILE=3 LWS_LAST8_TILE=2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3d_stack or batch_paths" 2>&1 | tail -3
