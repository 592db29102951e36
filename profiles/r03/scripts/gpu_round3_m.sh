set -x
O=gpurun_out/r3m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "refine or forward_bitexact or schedule_options or full_size or batch8 or large_batch or odd_size or noise_floor or literal" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for sp in 0 1; do
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt conv64_split=$sp > $O/bench_b1_split$sp.json 2>/dev/null
python bench.py --batch 2 --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt conv64_split=$sp > $O/bench_b2_split$sp.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt conv64_split=$sp > $O/bench_b8_split$sp.json 2>/dev/null
python bench.py --batch 4 --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined --opt conv64_split=$sp > $O/bench_b4_split$sp.json 2>/dev/null
done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_b1_pool.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); k=d['kernels']
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], k['ref_conv64'], (d.get('pipelined') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
"
