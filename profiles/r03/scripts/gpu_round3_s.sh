set -x
O=gpurun_out/r3s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "conv3d_stack or schedule_options or disparity_stages or full_size or batch8 or large_batch or odd or forward_bitexact or config3 or config5" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python tools/sbench.py --batch 1 > $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 >> $O/sbench.txt 2>&1
grep stage $O/sbench.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined > $O/bench_b1.json 2>/dev/null
python bench.py --batch 2 --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined > $O/bench_b2.json 2>/dev/null
python bench.py --batch 4 --steps 50 --warmup 10 --no-cpu-baseline --no-pipelined > $O/bench_b4.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_b8.json 2>/dev/null
python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined > $O/bench_kitti.json 2>/dev/null
python bench.py --size 544x960 --maxdisp0 32 --feature-fp16 --no-cpu-baseline --no-pipelined --steps 20 > $O/bench_cfg5.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['secondary']['stage2']['avg_launch_us'], d['secondary']['stage3']['avg_launch_us'], d['secondary']['kernel'])
    except Exception as e: print(f, 'ERR', e)
"
