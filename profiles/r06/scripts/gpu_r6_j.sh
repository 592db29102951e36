#!/bin/bash
# round 6, call j: roofline.traffic measured in the run (two rocprofv3 --pmc child passes started by bench.py itself before it
# touches the GPU): the driver's command line, the default line, batch 8; how long the passes take; the committed number beside it.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6j
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
show() { python -c "
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=d['roofline']
print('$1'.split('/')[-1], d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], 'in_run', r['traffic_measured_in_run'], 'committed', r.get('traffic_committed'), r.get('traffic_counters_kb'), r.get('traffic_in_run_note'), r.get('traffic_source'))"; }
( time python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_b1_driver_flags.json" 2> "$O/b1.err" ) 2>&1 | grep real; show "$O/bench_b1_driver_flags.json"
( time python bench.py > "$O/bench_b1.json" 2> "$O/b1d.err" ) 2>&1 | grep real; show "$O/bench_b1.json"
( time python bench.py --batch 8 --no-cpu-baseline --no-pipelined --steps 30 > "$O/bench_b8.json" 2> "$O/b8.err" ) 2>&1 | grep real; show "$O/bench_b8.json"
( time python bench.py --gpus 1 --steps 20 --warmup 5 --no-measure-traffic > "$O/bench_b1_no_measure.json" 2> /dev/null ) 2>&1 | grep real; show "$O/bench_b1_no_measure.json"
tail -3 "$O/b1.err"
