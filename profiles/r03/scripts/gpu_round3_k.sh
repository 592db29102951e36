set -x
O=gpurun_out/r3k; mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29540 tools/gather_probe.py > $O/gather_probe.txt 2>&1; grep pairs $O/gather_probe.txt
p=29541
for steps in 60 200; do for gp in 1 4 8 16 32; do
p=$((p+1))
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 1 --steps $steps --warmup 10 --no-cpu-baseline --gather-pairs $gp > $O/bench_tr1_s${steps}_g$gp.json 2>/dev/null
done; done
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); c=d['collective']
        print(f, d['value'], d['ms_per_step'], c['gather_every_steps'], c['gathers_in_timed_region'], c['ms_per_step_without_gather'], c['overhead_pct'])
    except Exception as e: print(f, 'ERR', e)
"
