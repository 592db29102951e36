#!/bin/bash
# Round 4, call C: the LDS-window k_volume_l1_warp (parity, per-launch times, in-forward effect) and the root-rank gather
# budget with RCCL's SendRecv kernel actually running (ncclSend/ncclRecv to self: tools/gather_probe.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "volume_warp or schedule_options or disparity_stages or full_size" > "$O/pytest_warp.txt" 2>&1; tail -5 "$O/pytest_warp.txt"
for cfg in "b1:--batch 1" "b8:--batch 8" "b8_368x1232:--batch 8 --size 368x1232" "b1_368x1232:--batch 1 --size 368x1232"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  python tools/wbench.py $args >> "$O/wbench.txt" 2>> "$O/wbench.err"
  python tools/wbench.py $args --noise 6 >> "$O/wbench_noise6.txt" 2>> "$O/wbench.err"
done
cat "$O/wbench.txt"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/wkt" -o run -- python3 "$R/tools/wbench.py" --batch 8 --size 368x1232 --iters 50 > /dev/null 2>&1
cd "$R"; grep -h "k_volume_l1_warp" "$O"/wkt/*kernel_stats.csv > "$O/wbench_b8_368x1232_kernel_stats.csv"; rm -rf "$O/wkt"; cat "$O/wbench_b8_368x1232_kernel_stats.csv"
for f in 1 0; do
  python bench.py --no-cpu-baseline --no-pipelined --steps 200 --opt warp_form=$f > "$O/bench_b1_warp_form$f.json" 2> /dev/null
  python bench.py --no-cpu-baseline --no-pipelined --batch 8 --steps 30 --opt warp_form=$f > "$O/bench_b8_warp_form$f.json" 2> /dev/null
  python bench.py --no-cpu-baseline --no-pipelined --batch 8 --size 368x1232 --steps 10 --warmup 3 --opt warp_form=$f > "$O/bench_cfg3_warp_form$f.json" 2> /dev/null
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['hbm_kernels'].get('volume_l1_warp'))
    except Exception as e: print(f, 'ERR', e)
" | tee "$O/bench_warp_summary.txt"
python tools/stamps.py warp2 1 > "$O/stamps_warp2_b1.txt" 2>/dev/null; python tools/stamps.py warp3 8 > "$O/stamps_warp3_b8.txt" 2>/dev/null
python -m lwsnet_amd.build --force > /dev/null 2>&1
cat "$O/stamps_warp2_b1.txt"
# ---- gather budget, RCCL kernel running
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
port=29700
run() {
  port=$((port + 1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $port tools/gather_probe.py --beside "${@:2}" 2>> "$O/err.txt" | grep '^{' | sed "s/^/$1 /" >> "$O/gather_root_emulation.txt"
}
for cap in default 1 2 4 8; do
  if [ "$cap" = default ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS
  else export NCCL_MAX_NCHANNELS=$cap NCCL_MIN_NCHANNELS=$cap NCCL_MAX_P2P_NCHANNELS=$cap NCCL_MIN_P2P_NCHANNELS=$cap; fi
  run "cap=$cap" --batch 1 --emulate-world 8 --gather-pairs 8,16,32 --steps 384
  run "cap=$cap" --batch 8 --emulate-world 8 --gather-pairs 8,16 --steps 64
done
unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS
run "selfcopy" --batch 1 --emulate-world 8 --gather-pairs 8 --steps 384 --self-copy
python - "$O/gather_root_emulation.txt" <<'PY' | tee "$O/gather_root_emulation_table.txt"
import sys, json
for l in open(sys.argv[1]):
    lab, js = l.split(' ', 1); d = json.loads(js)
    print(f"{lab:11s} B={d['batch']} emu_world={d['emulated_world']} pairs/gather={d['pairs_per_rank_per_gather']:3d} root_MB={d['MB_written_on_root_per_gather']:7.1f} plain={d['ms_per_step_plain']:.4f} slots={d['ms_per_step_slots_only']:.4f} with={d['ms_per_step_with_gather']:.4f} overhead={d['overhead_pct']:6.2f}% vs_slots={d['overhead_pct_vs_slots_only']:6.2f}% {d['overhead_pct_min_max']}")
PY
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
cd /tmp
for cap in default 2; do
  if [ "$cap" = default ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS
  else export NCCL_MAX_NCHANNELS=$cap NCCL_MIN_NCHANNELS=$cap NCCL_MAX_P2P_NCHANNELS=$cap NCCL_MIN_P2P_NCHANNELS=$cap; fi
  export MASTER_PORT=$((29791 + ${#cap}))
  rocprofv3 --kernel-trace --output-format csv -d "$O/gt_$cap" -o run -- python3 "$R/tools/gather_probe.py" --beside --batch 1 --emulate-world 8 --gather-pairs 8 --steps 128 --reps 1 > "$O/gt_$cap.json" 2> "$O/gt_$cap.err"
  python3 "$R/tools/gather_trace.py" $(find "$O/gt_$cap" -name '*kernel_trace.csv' | head -1) > "$O/gather_trace_b1_cap_$cap.txt" 2>&1
  rm -rf "$O/gt_$cap"
done
cd "$R"
head -30 "$O/gather_trace_b1_cap_default.txt"
du -sh "$O"
