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
python tools/sbench.py --batch 8 --size 368x1232 > "$O/sbench_b8_368x1232.txt" 2> /dev/null
# lws_pool throughput matrix, the collective's cost under a world-of-one torchrun, and the two micro-benchmarks DESIGN.md cites
GPU_MAX_HW_QUEUES=8 python tools/pool_bench.py --workers 2,3,4,6 > "$O/pool_bench_q8.txt" 2> /dev/null
GPU_MAX_HW_QUEUES=4 python tools/pool_bench.py --workers 3,4 > "$O/pool_bench_q4.txt" 2> /dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > "$O/bench_torchrun_world1_b1.json" 2> /dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --batch 8 --steps 40 --warmup 5 --no-cpu-baseline > "$O/bench_torchrun_world1_b8.json" 2> /dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > "$O/bench_b1_steps20.json" 2> /dev/null
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined > "$O/bench_b1_steps200.json" 2> /dev/null
(cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o mfma4x4_bcast mfma4x4_bcast.hip && ./mfma4x4_bcast > "$O/micro_mfma4x4_bcast.txt" 2>&1; hipcc --offload-arch=gfx950 -O3 -o copybw copybw.hip && ./copybw 8 > "$O/micro_copybw_b8.txt" 2>&1; ./copybw 1 > "$O/micro_copybw_b1.txt" 2>&1)
python tools/stamps.py mid8q3 8 > "$O/stamps_mid8q_stage3_b8.txt" 2> /dev/null
python tools/stamps.py mid8q3 1 > "$O/stamps_mid8q_stage3_b1.txt" 2> /dev/null
python tools/stamps.py mid8_3 8 > "$O/stamps_mid8_stage3_b8.txt" 2> /dev/null
python -m lwsnet_amd.build --force > /dev/null 2>&1
python tools/split_bf16_numerics.py --pairs 8 > "$O/split_bf16_numerics_64x256.txt" 2> /dev/null
python tools/split_bf16_numerics.py --pairs 6 --size 128x384 > "$O/split_bf16_numerics_128x384.txt" 2> /dev/null
# the opt-in numerics mode as whole steps (never the headline) and its refinement kernel
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt split_bf16=1 > "$O/bench_b1_split_bf16.json" 2> /dev/null
python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt split_bf16=1 > "$O/bench_b8_split_bf16.json" 2> /dev/null
python bench.py --size 544x960 --maxdisp0 32 --feature-fp16 --no-cpu-baseline --no-pipelined --steps 20 --opt split_bf16=1 > "$O/bench_cfg5_split_bf16.json" 2> /dev/null
python tools/rbench.py --opt conv64_form=1 > "$O/rbench_b1_conv64x.txt" 2> /dev/null
python tools/rbench.py --batch 8 --iters 20 --opt conv64_form=1 > "$O/rbench_b8_conv64x.txt" 2> /dev/null
(cd tools/micro && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o split_bf16 split_bf16.hip && ./split_bf16 > "$O/micro_split_bf16.txt" 2>&1)
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29535 tools/gather_probe.py 2> /dev/null | grep pairs > "$O/gather_probe.txt"
