O=gpurun_out/r3x; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -s -k "split_bf16 or conv3d_stack or schedule_options" > $O/pytest.txt 2>&1; grep -E "split-bf16|mean|passed|failed|Error" $O/pytest.txt | tail -30
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt split_bf16=1 > $O/bench_b1_all.json 2>/dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt split_bf16=1 > $O/bench_b8_all.json 2>/dev/null
python bench.py --size 544x960 --maxdisp0 32 --feature-fp16 --no-cpu-baseline --no-pipelined --steps 20 > $O/bench_cfg5.json 2>/dev/null
python bench.py --size 544x960 --maxdisp0 32 --feature-fp16 --no-cpu-baseline --no-pipelined --steps 20 --opt split_bf16=1 > $O/bench_cfg5_all.json 2>/dev/null
python bench.py --batch 8 --size 368x1232 --no-cpu-baseline --no-pipelined --steps 10 --warmup 3 --opt split_bf16=1 > $O/bench_cfg3_all.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['dtype'][:40])
    except Exception as e: print(f, 'ERR', e)
"
