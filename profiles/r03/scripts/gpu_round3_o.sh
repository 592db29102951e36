set -x
O=gpurun_out/r3o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "split_bf16 or conv3d_stack" > $O/pytest.txt 2>&1; grep -v "^$" $O/pytest.txt | tail -15
for f in 0 1; do
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid16_form=$f > $O/bench_b1_x$f.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt mid16_form=$f > $O/bench_b8_x$f.json 2>/dev/null
python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --opt mid16_form=$f > $O/bench_kitti_x$f.json 2>/dev/null
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels']
        print(f, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], k['conv3d_mid16'])
    except Exception as e: print(f, 'ERR', e)
"
