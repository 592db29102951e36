set -x
O=gpurun_out/r3t; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "split_bf16 or refine" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
for b in 1 8; do
  python tools/rbench.py --batch $b > $O/rbench_b$b.txt 2>&1
  python tools/rbench.py --batch $b --opt conv64_form=1 > $O/rbench_b${b}_x.txt 2>&1
  grep -E "ref_conv64|wall" $O/rbench_b$b.txt $O/rbench_b${b}_x.txt
done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt conv64_form=1 > $O/bench_b1_c64x.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt conv64_form=1 --opt mid16_form=1 > $O/bench_b1_both.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined > $O/bench_b1.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined > $O/bench_b8.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt conv64_form=1 --opt mid16_form=1 > $O/bench_b8_both.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
"
