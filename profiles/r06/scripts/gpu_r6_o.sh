#!/bin/bash
# round 6, call o: the supervised N-rank job at BASELINE config 4's launch shape -- 4 and 8 ranks of the real HIP path, all
# sharing this one GPU over gloo (RCCL refuses duplicate devices): supervisors, fresh workers, both legs (1 pair and 8 pairs
# per rank per step), every rank's gathered slot checked against the unsharded forward.  value is null by construction.
# Also the same through torchrun directly (how the driver starts it) and with a hung rank injected (watchdog + fallback).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6o
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
show() { python -c "
import json
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); c=d.get('collective') or {}; c4=d.get('config4') or {}
print('$1'.split('/')[-1], 'n', d['n_gpus'], 'value', d['value'], 'ms', d.get('ms_per_step'), 'all_equal', c.get('all_ranks_slots_equal_unsharded'), 'config4:', c4.get('global_batch'), c4.get('ms_per_step'), c4.get('all_ranks_slots_equal_unsharded'), 'clocks', (d.get('roofline') or {}).get('clock_ghz_per_rank'), 'attempts', d.get('attempts'), 'error', d.get('error'))"; }
for n in 4 8; do
  ( time python bench.py --gpus $n --one-gpu --steps 6 --warmup 2 --no-cpu-baseline > "$O/bench_${n}_ranks_one_gpu.json" 2> "$O/bench_${n}.err" ) 2>&1 | grep real
  show "$O/bench_${n}_ranks_one_gpu.json"
done
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 8 --one-gpu --steps 6 --warmup 2 --no-cpu-baseline > "$O/bench_8_ranks_one_gpu_torchrun.json" 2> "$O/bench_8t.err" ) 2>&1 | grep real
show "$O/bench_8_ranks_one_gpu_torchrun.json"
( time LWS_BENCH_INJECT=hang-collective:3 python bench.py --gpus 4 --one-gpu --steps 6 --warmup 2 --no-cpu-baseline --job-timeout 60 --init-timeout 20 > "$O/bench_4_ranks_hung_rank.json" 2> "$O/bench_4h.err" ) 2>&1 | grep real
show "$O/bench_4_ranks_hung_rank.json"
