set -x
O=gpurun_out/r3h; mkdir -p $O
for cfg in "8 256x512 30" "4 256x512 50" "8 368x1232 10"; do set -- $cfg
for opts in "--opt left_at=0" "--opt left_at=2" "--opt left_at=2 --opt split_heads=0" "--opt left_at=0 --opt split_heads=0"; do
n=$(echo $opts | tr -d ' -' | tr '=' '_')
python bench.py --batch $1 --size $2 --steps $3 --warmup 5 --no-cpu-baseline --no-pipelined $opts > $O/bench_b$1_$2_$n.json 2>/dev/null
done; done
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac'))
    except Exception as e: print(f, 'ERR', e)
"
