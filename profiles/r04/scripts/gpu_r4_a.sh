#!/bin/bash
# Round 4, call A (at r03's HEAD kernels): the evidence VERDICT r3 found missing + baselines for this round's kernel work.
#   gpurun --timeout 2400 -- 'bash profiles/r04/scripts/gpu_r4_a.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
python bench.py --no-cpu-baseline > "$O/bench_b1_before.json" 2> "$O/bench_b1_before.err"
# bench_repeats: 8 fresh processes, the driver's flags
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
cat "$O/bench_repeats.txt"
# timeline of one B=1 forward
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o run -- python3 "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined > "$O/kt_bench.json" 2> "$O/kt.err"
cd "$R"
python tools/timeline.py "$O/kt/run_kernel_trace.csv" 90 > "$O/timeline_b1_before.txt" 2>&1
cp "$O"/kt/run_kernel_stats.csv "$O/kernel_stats_b1_256x512_before.csv"; rm -rf "$O"/kt
# counters at B=8 256x512 and 8 x 368x1232 (separate passes, no trace domains)
cd /tmp
for cfg in "b8_256x512:--batch 8" "b8_368x1232:--batch 8 --size 368x1232"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_$tag" -o run -- python3 "$R/bench.py" $args --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write_$tag" -o run -- python3 "$R/bench.py" $args --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE \
    --output-format csv -d "$O/sq_$tag" -o run -- python3 "$R/bench.py" $args --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
  python3 "$R/tools/pmc_summary.py" "$O" "$tag" "python3 bench.py $args --steps 3 --warmup 2 (round-3 kernels, commit 2001c4a + tools)" "$O/fetch_$tag" "$O/write_$tag" "$O/sq_$tag"
  rm -rf "$O/fetch_$tag" "$O/write_$tag" "$O/sq_$tag"
done
cd "$R"
# refinement at config 3, chunks piped over two streams or not
python tools/rbench.py --batch 8 --size 368x1232 --iters 6 --opt ref_pipe=0 > "$O/rbench_b8_368x1232_pipe0.txt" 2> "$O/rbench_c3_0.err"
python tools/rbench.py --batch 8 --size 368x1232 --iters 6 --opt ref_pipe=1 > "$O/rbench_b8_368x1232_pipe1.txt" 2> "$O/rbench_c3_1.err"
python tools/rbench.py --batch 1 --size 368x1232 --iters 20 > "$O/rbench_b1_368x1232.txt" 2> /dev/null
# in-kernel clock (stamped diagnostic builds; each rebuilds the library)
for k in "mid16 1" "mid16 8" "mid8q3 1" "mid8q3 8" "conv64 1" "conv64 8" "dws 8"; do
  echo "== tools/stamps.py $k" >> "$O/stamps_inkernel_clock.txt"
  python tools/stamps.py $k >> "$O/stamps_inkernel_clock.txt" 2> /dev/null
done
python -m lwsnet_amd.build --force > /dev/null 2>&1
du -sh "$O"; grep -c . "$O/stamps_inkernel_clock.txt"
