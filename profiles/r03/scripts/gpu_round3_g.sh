set -x
O=gpurun_out/r3g; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for opt in "left_at=0" "left_at=2" "side_streams=0" "split_heads=0" "split_heads=1"; do
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt $opt > $O/bench_b8_$opt.json 2>/dev/null
done
GPU_MAX_HW_QUEUES=4 python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_b8_q4.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b8_default.json 2>/dev/null
python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined > $O/bench_kitti_default.json 2>/dev/null
python bench.py --batch 2 --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_b2_default.json 2>/dev/null
python bench.py --batch 4 --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_b4_default.json 2>/dev/null
python bench.py > $O/bench_b1_default.json 2>$O/bench_b1_default.err
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac'), (d.get('pipelined') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
"
