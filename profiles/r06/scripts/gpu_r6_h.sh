#!/bin/bash
# round 6, call h: two launch-plan knobs at the larger batches, two passes: the second fork one middle layer earlier
# (fork2_after = 3; round 5 measured +0.3-0.6 % from batch 2 up) and the refinement chunk size (36 / 72 / 144 MB).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6h
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
run() { # tag, bench args
  python bench.py --no-cpu-baseline --no-pipelined --no-measure-traffic $2 > "$O/$1.json" 2> "$O/$1.err"
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1])
    print('$1', d['value'], d['ms_per_step'], 'mid16', d['roofline']['avg_launch_us'], 'clk', d['roofline']['clock_ghz'], 'v/clk', round(d['value']/d['roofline']['clock_ghz'],1))
except Exception as e: print('$1 ERR', e, open('$O/$1.err').read()[-300:])"
}
for pass in 1 2; do
  for o in "" "--opt fork2_after=3" "--opt fork2_after=2" "--opt ref_chunk_mb=36" "--opt ref_chunk_mb=144" "--opt ref_chunk_mb=36 --opt ref_pipe=1"; do
    tag=$(echo "$o" | tr -d ' -' | tr '=' '_'); tag=${tag:-default}
    run "p${pass}_b8_$tag" "--batch 8 --steps 40 $o"
    run "p${pass}_b4_$tag" "--batch 4 --steps 60 $o"
    run "p${pass}_cfg3_$tag" "--batch 8 --size 368x1232 --steps 12 --warmup 3 $o"
  done
done 2>&1 | tee "$O/ab_plan_knobs_large_batches.txt"
