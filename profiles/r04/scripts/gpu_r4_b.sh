#!/bin/bash
# Round 4, call B: the root rank's gather budget on ONE GPU.  World of one under torchrun, the message
# sized as the root of an 8-rank job writes it (tools/gather_probe.py --beside --emulate-world 8), swept over RCCL's channel
# caps and the gather cadence; then one kernel trace naming the kernels the RCCL kernel delays.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4b
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
port=29600
run() {   # $1 = label, rest = probe args ; environment caps exported by the caller
  port=$((port + 1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $port tools/gather_probe.py --beside "${@:2}" 2>> "$O/err.txt" | grep '^{' | sed "s/^/$1 /" >> "$O/gather_root_emulation.txt"
}
for cap in default 1 2 4; do
  if [ "$cap" = default ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS
  else export NCCL_MAX_NCHANNELS=$cap NCCL_MIN_NCHANNELS=$cap NCCL_MAX_P2P_NCHANNELS=$cap NCCL_MIN_P2P_NCHANNELS=$cap; fi
  run "cap=$cap" --batch 1 --emulate-world 8 --gather-pairs 8,16,32,64 --steps 384
  run "cap=$cap" --batch 8 --emulate-world 8 --gather-pairs 8,16,32 --steps 64
  run "cap=$cap" --batch 1 --emulate-world 1 --gather-pairs 8 --steps 384
done
# p2p cap only (collective channel count untouched)
unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS
export NCCL_MAX_P2P_NCHANNELS=1
run "p2pcap=1" --batch 1 --emulate-world 8 --gather-pairs 8,32 --steps 384
run "p2pcap=1" --batch 8 --emulate-world 8 --gather-pairs 8 --steps 64
unset NCCL_MAX_P2P_NCHANNELS
cat "$O/gather_root_emulation.txt" | python -c "
import sys, json
for l in sys.stdin:
    lab, js = l.split(' ', 1); d = json.loads(js)
    print(f\"{lab:10s} B={d['batch']} emu_world={d['emulated_world']} pairs/gather={d['pairs_per_rank_per_gather']:3d} root_MB={d['MB_written_on_root_per_gather']:7.1f} plain={d['ms_per_step_plain']:.4f} with={d['ms_per_step_with_gather']:.4f} overhead={d['overhead_pct']:6.2f}% {d['overhead_pct_min_max']}\")
" | tee "$O/gather_root_emulation_table.txt"
# idle gathers (issue -> complete) with RCCL's defaults
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29690 tools/gather_probe.py 2> /dev/null | grep pairs > "$O/gather_probe_idle.txt"
# which kernels does the RCCL kernel delay?  program directly after `--`; RANK etc. exported in this shell
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29691
cd /tmp; export TMPDIR=/tmp
for cap in default 2; do
  if [ "$cap" = default ]; then unset NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS
  else export NCCL_MAX_NCHANNELS=$cap NCCL_MIN_NCHANNELS=$cap NCCL_MAX_P2P_NCHANNELS=$cap NCCL_MIN_P2P_NCHANNELS=$cap; fi
  MASTER_PORT=$((29691 + ${#cap}))
  rocprofv3 --kernel-trace --output-format csv -d "$O/gt_$cap" -o run -- python3 "$R/tools/gather_probe.py" --beside --batch 1 --emulate-world 8 --gather-pairs 8 --steps 128 --reps 1 > "$O/gt_$cap.json" 2> "$O/gt_$cap.err"
  python3 "$R/tools/gather_trace.py" $(find "$O/gt_$cap" -name '*kernel_trace.csv' | head -1) > "$O/gather_trace_b1_cap_$cap.txt" 2>&1
  rm -rf "$O/gt_$cap"
done
unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT NCCL_MAX_NCHANNELS NCCL_MIN_NCHANNELS NCCL_MAX_P2P_NCHANNELS NCCL_MIN_P2P_NCHANNELS
cd "$R"
head -40 "$O/gather_trace_b1_cap_default.txt"
du -sh "$O"
