#!/bin/bash
# round 6, call n: the 16 x 16-tile form of the two half-resolution feature-extractor pairs (k_conv2d_pair<..., 16>) on large
# grids: parity, then the A/B against the 8 x 8 form (LWS_PAIR_TILE = 8 / 16, a temporary switch) at B = 4, 8 and 8 x 368x1232,
# two passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6n
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "feature_extraction or full_size or batch_paths" > "$O/pytest_fe.txt" 2>&1; tail -5 "$O/pytest_fe.txt"
run() {
  env $2 python bench.py --no-cpu-baseline --no-pipelined --no-measure-traffic $3 > "$O/$1.json" 2> "$O/$1.err"
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); k=d['kernels']
    print('$1', d['value'], d['ms_per_step'], 'feature', k['feature_conv2d'], 'clk', d['roofline']['clock_ghz'], 'v/clk', round(d['value']/d['roofline']['clock_ghz'],1))
except Exception as e: print('$1 ERR', e, open('$O/$1.err').read()[-300:])"
}
for pass in 1 2 3; do
  for t in 8 16; do
    run "p${pass}_b8_t$t" "LWS_PAIR_TILE=$t" "--batch 8 --steps 40"
    run "p${pass}_b4_t$t" "LWS_PAIR_TILE=$t" "--batch 4 --steps 60"
    run "p${pass}_b2_t$t" "LWS_PAIR_TILE=$t" "--batch 2 --steps 100"
    run "p${pass}_cfg3_t$t" "LWS_PAIR_TILE=$t" "--batch 8 --size 368x1232 --steps 12 --warmup 3"
  done
done 2>&1 | tee "$O/ab_pair_tile.txt"
