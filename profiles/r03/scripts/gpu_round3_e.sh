set -x
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "refine or schedule_options or batch8 or large_batch" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for ord in 0 1; do
python tools/sbench.py --batch 1 --opt conv3d_order=$ord >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 --opt conv3d_order=$ord >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 --size 368x1232 --opt conv3d_order=$ord >> $O/sbench.txt 2>&1
done
grep stage $O/sbench.txt
python tools/rbench.py --batch 8 --iters 30 2>&1 | grep -A6 "ref_order=0" > $O/rbench_b8.txt; cat $O/rbench_b8.txt
python tools/rbench.py --batch 1 --iters 30 2>&1 | grep -A6 "ref_order=0" > $O/rbench_b1.txt; cat $O/rbench_b1.txt
for mb in 0 72 36; do
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt mid8_form=1 --opt ref_chunk_mb=$mb > $O/bench_b8_chunk$mb.json 2>/dev/null
python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --opt mid8_form=1 --opt ref_chunk_mb=$mb > $O/bench_kitti_chunk$mb.json 2>/dev/null
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        k=d['kernels']
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], {n: k[n]['avg_us'] for n in ('conv3d_mid16','conv3d_mid8','ref_first','ref_dws','ref_conv64','ref_last')})
    except Exception as e: print(f, 'ERR', e)
"
