set -x
O=gpurun_out/r3l; mkdir -p $O
for rep in 1 2 3; do for sp in 0 0.3 1.0 2.5; do
sleep 4
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --spinup $sp > $O/bench_s20_sp${sp}_r$rep.json 2>/dev/null
done; done
sleep 4
python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_s200.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
"
