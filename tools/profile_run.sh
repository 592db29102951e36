#!/bin/bash
# Produces everything profiles/rNN is built from, on the GPU box, under gpurun_out/final:
#   gpurun --timeout 1500 -- 'bash tools/profile_run.sh'      then here:  python tools/collect_profiles.py
# Counter passes (--pmc) are separate rocprofv3 runs without any trace domain; the program after `--` is python3 itself.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1
python bench.py > "$O/bench_b1.json" 2> "$O/bench_b1.err"
python bench.py --batch 8 --no-cpu-baseline --no-pipelined --steps 30 > "$O/bench_b8.json" 2> /dev/null
python bench.py --batch 8 --size 368x1232 --no-cpu-baseline --no-pipelined --steps 10 --warmup 3 > "$O/bench_cfg3.json" 2> /dev/null
python bench.py --size 544x960 --maxdisp0 32 --feature-fp16 --no-cpu-baseline --no-pipelined --steps 20 > "$O/bench_cfg5.json" 2> /dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o run -- python3 "$R/bench.py" --steps 30 --warmup 10 --no-cpu-baseline --no-pipelined > "$O/kt_bench.json" 2> "$O/kt.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o run -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -o run -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE \
    --output-format csv -d "$O/sq" -o run -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
cd "$R"
# keep the merge-back small: the kernel trace itself is large and not needed
rm -f "$O"/kt/run_kernel_trace.csv
tail -c 400 "$O/bench_b1.json"; echo; tail -3 "$O/smoke.log"
# batch-8 kernel statistics (BASELINE configs 4 / 3 per-GPU workloads)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt8" -o run -- python3 "$R/bench.py" --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ktc3" -o run -- python3 "$R/bench.py" --batch 8 --size 368x1232 --steps 8 --warmup 3 --no-cpu-baseline --no-pipelined > /dev/null 2>&1
rm -f "$O"/kt8/run_kernel_trace.csv "$O"/ktc3/run_kernel_trace.csv
cd "$R"
python tools/sbench.py > "$O/sbench_b1.txt" 2> /dev/null
python tools/sbench.py --batch 8 > "$O/sbench_b8.txt" 2> /dev/null
python tools/rbench.py > "$O/rbench_b1.txt" 2> /dev/null
python tools/rbench.py --batch 8 --iters 20 > "$O/rbench_b8.txt" 2> /dev/null
