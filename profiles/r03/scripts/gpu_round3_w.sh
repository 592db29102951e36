set -x
O=gpurun_out/r3w; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "split_bf16 or options_validate" > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt
python tools/sbench.py --batch 1 > $O/sbench_b1.txt 2>&1
python tools/sbench.py --batch 8 > $O/sbench_b8.txt 2>&1
grep stage $O/sbench_b1.txt $O/sbench_b8.txt
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_form=2 > $O/bench_b1_mid8x.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_form=2 --opt conv64_form=1 --opt mid16_form=1 > $O/bench_b1_all.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt mid8_form=2 --opt conv64_form=1 --opt mid16_form=1 > $O/bench_b8_all.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
"
