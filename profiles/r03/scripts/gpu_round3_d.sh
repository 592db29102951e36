set -x
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "conv3d_stack or schedule_options or disparity_stages" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for ord in 0 1; do
python tools/sbench.py --batch 1 --opt conv3d_order=$ord >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 --opt conv3d_order=$ord >> $O/sbench.txt 2>&1
python tools/sbench.py --batch 8 --size 368x1232 --opt conv3d_order=$ord >> $O/sbench.txt 2>&1
done
grep stage $O/sbench.txt
for ord in 0 1; do for f in 0 1; do
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt conv3d_order=$ord --opt mid8_form=$f > $O/bench_b1_o${ord}_f$f.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt conv3d_order=$ord --opt mid8_form=$f > $O/bench_b8_o${ord}_f$f.json 2>/dev/null
done; done
python tools/stamps.py mid8q3 8 > $O/stamps_mid8q3_b8.txt 2>&1; cat $O/stamps_*.txt
python -m lwsnet_amd.build --force > /dev/null 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        k=d['kernels']
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], {n: k[n]['avg_us'] for n in ('conv3d_first','conv3d_mid16','conv3d_mid8','conv3d_last','ref_dws','ref_conv64')})
    except Exception as e: print(f, 'ERR', e)
"
