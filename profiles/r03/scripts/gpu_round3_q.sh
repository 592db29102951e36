set -x
O=gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "split_bf16 or conv3d_stack or schedule" > $O/pytest.txt 2>&1; grep -v "^$" $O/pytest.txt | grep "mean\|max \|passed\|failed\|split-bf16 stack" | tail -12
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_b1.json 2>$O/bench_b1.err
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_b8.json 2>/dev/null
python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_kitti.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], (d.get('pipelined') or {}).get('value'), d.get('split_bf16'))
    except Exception as e: print(f, 'ERR', e)
"
tail -3 $O/bench_b1.err
