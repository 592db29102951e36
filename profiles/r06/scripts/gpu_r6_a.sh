#!/bin/bash
# round 6, call a: the GPU suite at HEAD (ABI v8: pruned options, interp_align_mode, graph capture, fused refinement1_left head,
# supervised bench jobs), default bench lines incl. clock_ghz, the torchrun / one-GPU / fallback launch shapes, and the A/B
# of "fuse_first" bit 1 (refinement1_left's 3 -> 32 convolution inside its first block) at B = 1 / 8 / 8 x 368x1232.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
timeout 3000 python -m pytest tests -x -q -m gpu > "$O/pytest.txt" 2>&1; tail -15 "$O/pytest.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_b1_driver_flags.json" 2> "$O/bench_b1.err"; tail -c 300 "$O/bench_b1_driver_flags.json"
python bench.py --gpus 2 --one-gpu --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_two_ranks_one_gpu.json" 2> "$O/bench_two_ranks.err"; tail -c 900 "$O/bench_two_ranks_one_gpu.json"
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --config4 > "$O/bench_torchrun_world1_config4.json" 2> "$O/bench_torchrun.err"; tail -c 900 "$O/bench_torchrun_world1_config4.json"
for ff in 1 3; do
  python bench.py --no-cpu-baseline --no-pipelined --steps 200 --warmup 20 --opt fuse_first=$ff > "$O/ab_fuse_first${ff}_b1.json" 2> /dev/null
  python bench.py --batch 8 --no-cpu-baseline --no-pipelined --steps 40 --opt fuse_first=$ff > "$O/ab_fuse_first${ff}_b8.json" 2> /dev/null
  python bench.py --batch 8 --size 368x1232 --no-cpu-baseline --no-pipelined --steps 12 --warmup 3 --opt fuse_first=$ff > "$O/ab_fuse_first${ff}_cfg3.json" 2> /dev/null
done
for ff in 1 3; do
  python bench.py --no-cpu-baseline --no-pipelined --steps 200 --warmup 20 --opt fuse_first=$ff > "$O/ab2_fuse_first${ff}_b1.json" 2> /dev/null
  python bench.py --batch 8 --no-cpu-baseline --no-pipelined --steps 40 --opt fuse_first=$ff > "$O/ab2_fuse_first${ff}_b8.json" 2> /dev/null
  python bench.py --batch 8 --size 368x1232 --no-cpu-baseline --no-pipelined --steps 12 --warmup 3 --opt fuse_first=$ff > "$O/ab2_fuse_first${ff}_cfg3.json" 2> /dev/null
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/ab*_fuse_first*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        k = d['kernels']
        print(os.path.basename(f), d['value'], d['ms_per_step'], 'mid16', d['roofline']['avg_launch_us'], 'clk', d['roofline'].get('clock_ghz'),
              'ref_first', k.get('ref_first'), 'ref_dws', k.get('ref_dws'))
    except Exception as e:
        print(f, 'ERR', e)
PY
du -sh "$O"
