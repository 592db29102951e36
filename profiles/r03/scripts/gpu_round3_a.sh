set -x
mkdir -p gpurun_out/r3a
cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o mfma4x4_bcast mfma4x4_bcast.hip && ./mfma4x4_bcast > ../../gpurun_out/r3a/micro.txt 2>&1; cd ../..
cat gpurun_out/r3a/micro.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.txt 2>&1; tail -15 gpurun_out/r3a/pytest.txt
python tools/sbench.py --batch 1 > gpurun_out/r3a/sbench_b1.txt 2>&1
python tools/sbench.py --batch 8 > gpurun_out/r3a/sbench_b8.txt 2>&1
python tools/sbench.py --batch 8 --size 368x1232 > gpurun_out/r3a/sbench_b8_kitti.txt 2>&1
python tools/rbench.py --batch 1 > gpurun_out/r3a/rbench_b1.txt 2>&1
python tools/rbench.py --batch 8 > gpurun_out/r3a/rbench_b8.txt 2>&1
cat gpurun_out/r3a/sbench_*.txt gpurun_out/r3a/rbench_*.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench_s20.json 2> gpurun_out/r3a/bench_s20.err
python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r3a/bench_s200.json 2> gpurun_out/r3a/bench_s200.err
for o in 0 1 2; do python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt ref_order=$o > gpurun_out/r3a/bench_b8_ro$o.json 2>&1; done
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt mid8_form=1 > gpurun_out/r3a/bench_b8_q.json 2>&1
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_form=1 > gpurun_out/r3a/bench_b1_q.json 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline'], d.get('secondary'), d.get('pipelined'))
    except Exception as e: print(f, 'ERR', e)
"
