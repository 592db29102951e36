#!/bin/bash
# round 6, call e: (1) the I/O kernels and the pipelined CLI that uses them; (2) the e2e table again; (3) why the profile run's
# `torchrun --nproc-per-node 1 bench.py --batch 8` read 20.9 % gather overhead (r03-r05: 0.2-0.7 %): the same job three times
# through the supervisor and three times with the rank process as the worker itself (LWS_BENCH_WORKER=1).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6e
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cli_directory or io_kernels or config1" > "$O/pytest_io.txt" 2>&1; tail -6 "$O/pytest_io.txt"
timeout 1500 python tools/e2e_cli.py --pairs 200 --workers 1 4 8 12 16 24 > "$O/e2e_cli.txt" 2> "$O/e2e_cli.err"; cat "$O/e2e_cli.txt"; tail -3 "$O/e2e_cli.err"
summ() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); c=d['collective']
print('$1'.split('/')[-1], d['value'], d['ms_per_step'], 'without', c.get('ms_per_step_without_gather'), 'overhead', c.get('overhead_pct'), 'clk', d['roofline'].get('clock_ghz'))"; }
for i in 1 2 3; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2954$i bench.py --gpus 1 --batch 8 --steps 40 --warmup 5 --no-cpu-baseline > "$O/tr_b8_supervised_$i.json" 2> /dev/null; summ "$O/tr_b8_supervised_$i.json"
  LWS_BENCH_WORKER=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2955$i bench.py --gpus 1 --batch 8 --steps 40 --warmup 5 --no-cpu-baseline > "$O/tr_b8_direct_$i.json" 2> /dev/null; summ "$O/tr_b8_direct_$i.json"
done
for i in 1 2; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2956$i bench.py --gpus 1 --batch 8 --steps 200 --warmup 5 --no-cpu-baseline > "$O/tr_b8_supervised_200_$i.json" 2> /dev/null; summ "$O/tr_b8_supervised_200_$i.json"
done
