#!/bin/bash
# Round 4, call D: k_ref_dws on packed float32 math + the FIRST form at four workgroups per CU (parity, per-launch times),
# and the gather emulation again with the host cost of the call and a timeline around one RCCL kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4d
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "refine or schedule_options or full_size or forward_batch" > "$O/pytest_refine.txt" 2>&1; tail -5 "$O/pytest_refine.txt"
python tools/rbench.py > "$O/rbench_b1.txt" 2>/dev/null
python tools/rbench.py --batch 8 --iters 20 > "$O/rbench_b8.txt" 2>/dev/null
python tools/rbench.py --batch 8 --size 368x1232 --iters 6 > "$O/rbench_b8_368x1232.txt" 2>/dev/null
head -6 "$O/rbench_b1.txt"; head -6 "$O/rbench_b8.txt"; head -6 "$O/rbench_b8_368x1232.txt"
python bench.py --no-cpu-baseline --no-pipelined --steps 200 > "$O/bench_b1.json" 2> /dev/null
python bench.py --no-cpu-baseline --no-pipelined --batch 8 --steps 30 > "$O/bench_b8.json" 2> /dev/null
python bench.py --no-cpu-baseline --no-pipelined --batch 8 --size 368x1232 --steps 10 --warmup 3 > "$O/bench_cfg3.json" 2> /dev/null
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['hbm_kernels'].get('ref_dws'))
    except Exception as e: print(f, 'ERR', e)
" | tee "$O/bench_summary.txt"
for cfg in "b1:--batch 1" "b8:--batch 8" "b8_368x1232:--batch 8 --size 368x1232"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  python tools/wbench.py $args >> "$O/wbench.txt" 2>> "$O/wbench.err"
done
cat "$O/wbench.txt"
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
port=29800
run() {
  port=$((port + 1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $port tools/gather_probe.py --beside "${@:2}" 2>> "$O/err.txt" | grep '^{' | sed "s/^/$1 /" >> "$O/gather_root_emulation.txt"
}
run "7pairs" --batch 1 --emulate-world 8 --gather-pairs 8,16 --steps 384
run "oneop" --batch 1 --emulate-world 8 --gather-pairs 8,16 --steps 384 --one-op
run "oneop" --batch 8 --emulate-world 8 --gather-pairs 8,16 --steps 64 --one-op
run "7pairs" --batch 8 --emulate-world 8 --gather-pairs 8,16 --steps 64
export GPU_MAX_HW_QUEUES=16
run "q16_7pairs" --batch 1 --emulate-world 8 --gather-pairs 8 --steps 384
run "q16_oneop" --batch 8 --emulate-world 8 --gather-pairs 8 --steps 64 --one-op
unset GPU_MAX_HW_QUEUES
python - "$O/gather_root_emulation.txt" <<'PY' | tee "$O/gather_root_emulation_table.txt"
import sys, json
for l in open(sys.argv[1]):
    lab, js = l.split(' ', 1); d = json.loads(js)
    print(f"{lab:11s} B={d['batch']} pairs/gather={d['pairs_per_rank_per_gather']:3d} root_MB={d['MB_written_on_root_per_gather']:7.1f} plain={d['ms_per_step_plain']:.4f} slots={d['ms_per_step_slots_only']:.4f} with={d['ms_per_step_with_gather']:.4f} overhead={d['overhead_pct']:6.2f}% host_us/gather={d['host_us_per_gather_call']} {d['overhead_pct_min_max']}")
PY
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29899
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/gt" -o run -- python3 "$R/tools/gather_probe.py" --beside --batch 1 --emulate-world 8 --gather-pairs 8 --steps 128 --reps 1 > "$O/gt.json" 2> "$O/gt.err"
python3 "$R/tools/gather_trace.py" $(find "$O/gt" -name '*kernel_trace.csv' | head -1) > "$O/gather_trace_b1.txt" 2>&1
rm -rf "$O/gt"
export MASTER_PORT=29898
rocprofv3 --kernel-trace --output-format csv -d "$O/gt8" -o run -- python3 "$R/tools/gather_probe.py" --beside --batch 8 --emulate-world 8 --gather-pairs 8 --steps 32 --reps 1 --one-op > "$O/gt8.json" 2> "$O/gt8.err"
python3 "$R/tools/gather_trace.py" $(find "$O/gt8" -name '*kernel_trace.csv' | head -1) > "$O/gather_trace_b8_oneop.txt" 2>&1
rm -rf "$O/gt8"
cd "$R"
head -70 "$O/gather_trace_b1.txt"
du -sh "$O"
