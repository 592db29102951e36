#!/bin/bash
# round 5, call g: the driver's sequence at HEAD (build, pytest -m gpu, smoke, bench with the driver's flags) + default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5g
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 2700 python -m pytest tests -x -q -m gpu > "$O/pytest.txt" 2>&1; tail -4 "$O/pytest.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_b1_driver_flags.json" 2> "$O/bench.err"
python bench.py > "$O/bench_b1.json" 2>> "$O/bench.err"
python -c "
import json
for f in ('$O/bench_b1_driver_flags.json','$O/bench_b1.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('step_frac'), (d.get('pipelined') or {}).get('value'), ((d.get('pipelined') or {}).get('roofline') or {}).get('frac'), d['cpu_baseline']['value'])
    print(d['roofline'].get('traffic_note'))
"
