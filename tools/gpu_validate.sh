# What the driver runs at round end, at HEAD: build check, GPU tests, smoke, default bench (--steps 20 --warmup 5 as in round 2's driver run).
set -x
O=gpurun_out/validate; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $O/build.log 2>&1; tail -1 $O/build.log
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json
for f in ('$O/bench.json','$O/bench_default.json'):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('step_frac'), (d.get('pipelined') or {}).get('value'), d['cpu_baseline']['value'])
"
