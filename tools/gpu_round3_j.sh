set -x
O=gpurun_out/r3j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "schedule_options or batch8 or large_batch or batch_paths or profiler or pool or config3" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for sp in 0 1; do
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt split_batch=$sp > $O/bench_b8_split$sp.json 2>/dev/null
python bench.py --batch 4 --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined --opt split_batch=$sp > $O/bench_b4_split$sp.json 2>/dev/null
python bench.py --batch 2 --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined --opt split_batch=$sp > $O/bench_b2_split$sp.json 2>/dev/null
python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --opt split_batch=$sp > $O/bench_kitti_split$sp.json 2>/dev/null
done
python bench.py --batch 16 --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined > $O/bench_b16.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('pairs_per_launch'), d['roofline'].get('step_frac'), (d.get('secondary') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
"
