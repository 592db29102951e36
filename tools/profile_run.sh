#!/bin/bash
# Produces everything profiles/rNN is built from, on the GPU box, under gpurun_out/final (summaries only: the raw rocprofv3
# CSVs are condensed on the box by tools/pmc_summary.py / tools/timeline.py and deleted -- gpurun merges back <= 64 MiB):
#   gpurun --timeout 2400 -- 'bash tools/profile_run.sh'      then here:  python tools/collect_profiles.py gpurun_out/final profiles/r06
# Counter passes (--pmc) are separate rocprofv3 runs without any trace domain; the program after `--` is python3 itself.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1
python bench.py > "$O/bench_b1.json" 2> "$O/bench_b1.err"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_b1_driver_flags.json" 2> /dev/null
python bench.py --batch 8 --no-cpu-baseline --no-pipelined --steps 30 > "$O/bench_b8.json" 2> /dev/null
python bench.py --batch 8 --size 368x1232 --no-cpu-baseline --no-pipelined --steps 10 --warmup 3 > "$O/bench_cfg3.json" 2> /dev/null
python bench.py --size 544x960 --maxdisp0 32 --feature-fp16 --no-cpu-baseline --no-pipelined --steps 20 > "$O/bench_cfg5.json" 2> /dev/null
# the N-rank code path with two processes on this one GPU (gloo; value is null: the ranks share the chip): both legs -- 1 pair per
# GPU per step, and config 4's 8 pairs per GPU per step
python bench.py --gpus 2 --one-gpu --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_two_ranks_one_gpu.json" 2> /dev/null
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('run $i --steps 20 --warmup 5:', d['value'], 'pairs/s', d['ms_per_step'], 'ms', 'mid16', d['roofline']['avg_launch_us'], 'us')" >> "$O/bench_repeats.txt"
done
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('run $i --steps 200 --warmup 10:', d['value'], 'pairs/s', d['ms_per_step'], 'ms', 'mid16', d['roofline']['avg_launch_us'], 'us')" >> "$O/bench_repeats.txt"
done
cd /tmp; export TMPDIR=/tmp
# kernel statistics + the timeline of one B=1 forward (back = 50 latency + 10 breakdown forwards behind the timed region + 30)
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o run -- python3 "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined > "$O/bench_under_rocprof.json" 2> "$O/kt.err"
python3 "$R/tools/timeline.py" "$O/kt/run_kernel_trace.csv" 90 > "$O/timeline_b1.txt" 2>&1
python3 "$R/tools/overlap_account.py" "$O/kt/run_kernel_trace.csv" 90 8 > "$O/overlap_account_b1_256x512.txt" 2>&1
cp "$O/kt/run_kernel_stats.csv" "$O/kernel_stats_b1_256x512.csv"; rm -rf "$O/kt"
# B=8 and 8 x 368x1232: kernel statistics, the timeline of ONE forward from the middle of the timed region (the forwards behind it
# are the breakdown pass, event-bracketed, and the latency pass, one forward per host synchronisation: 60 forwards from the end),
# and the overlap account of the side stream (tools/overlap_account.py)
for cfg in "b8_256x512:--batch 8 --steps 20 --warmup 5:70" "b8_368x1232:--batch 8 --size 368x1232 --steps 8 --warmup 3:64"; do
  tag=${cfg%%:*}; rest=${cfg#*:}; args=${rest%%:*}; back=${rest##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt_$tag" -o run -- python3 "$R/bench.py" $args --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  cp "$O/kt_$tag/run_kernel_stats.csv" "$O/kernel_stats_$tag.csv"
  python3 "$R/tools/timeline.py" "$O/kt_$tag/run_kernel_trace.csv" $back > "$O/timeline_${tag}.txt" 2>&1
  python3 "$R/tools/overlap_account.py" "$O/kt_$tag/run_kernel_trace.csv" $back 4 > "$O/overlap_account_${tag}.txt" 2>&1
  rm -rf "$O/kt_$tag"
done
# counters: FETCH_SIZE, WRITE_SIZE and the SQ set in separate passes, for the three published workloads
for cfg in "b1_256x512:" "b8_256x512:--batch 8" "b8_368x1232:--batch 8 --size 368x1232"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_$tag" -o run -- python3 "$R/bench.py" $args --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write_$tag" -o run -- python3 "$R/bench.py" $args --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE \
    --output-format csv -d "$O/sq_$tag" -o run -- python3 "$R/bench.py" $args --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  python3 "$R/tools/pmc_summary.py" "$O" "$tag" "python3 bench.py $args --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined" "$O/fetch_$tag" "$O/write_$tag" "$O/sq_$tag"
  rm -rf "$O/fetch_$tag" "$O/write_$tag" "$O/sq_$tag"
done
cd "$R"
python tools/sbench.py > "$O/sbench_b1.txt" 2> /dev/null
python tools/sbench.py --batch 8 > "$O/sbench_b8.txt" 2> /dev/null
python tools/sbench.py --batch 8 --size 368x1232 > "$O/sbench_b8_368x1232.txt" 2> /dev/null
python tools/rbench.py > "$O/rbench_b1.txt" 2> /dev/null
python tools/rbench.py --batch 8 --iters 20 > "$O/rbench_b8.txt" 2> /dev/null
python tools/rbench.py --batch 8 --size 368x1232 --iters 6 > "$O/rbench_b8_368x1232.txt" 2> /dev/null
for cfg in "--batch 1" "--batch 8" "--batch 8 --size 368x1232"; do python tools/wbench.py $cfg >> "$O/wbench.txt" 2> /dev/null; done
GPU_MAX_HW_QUEUES=8 python tools/pool_bench.py --workers 2,3,4,6 > "$O/pool_bench_q8.txt" 2> /dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --config4 > "$O/bench_torchrun_world1_b1.json" 2> /dev/null
LWS_BENCH_INJECT=init-fail:0 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --config4 > "$O/bench_torchrun_world1_fallback_gloo_host.json" 2> /dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --batch 8 --steps 40 --warmup 5 --no-cpu-baseline > "$O/bench_torchrun_world1_b8.json" 2> /dev/null
# in-kernel clock and phase stamps (stamped diagnostic builds; each rebuilds the library)
for k in "mid16 1" "mid16 8" "mid8q3 1" "mid8q3 8" "conv64 1" "conv64 8" "dws 8" "warp3 8"; do
  echo "== tools/stamps.py $k" >> "$O/stamps_inkernel_clock.txt"
  python tools/stamps.py $k >> "$O/stamps_inkernel_clock.txt" 2> /dev/null
done
python -m lwsnet_amd.build --force > /dev/null 2>&1
# the opt-in numerics mode as whole steps (never the headline)
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt split_bf16=7 > "$O/bench_b1_split_bf16.json" 2> /dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt split_bf16=7 > "$O/bench_b8_split_bf16.json" 2> /dev/null
tail -c 300 "$O/bench_b1.json"; echo; tail -2 "$O/smoke.log"; cat "$O/bench_repeats.txt"; du -sh "$O"
