#!/bin/bash
# round 5, call c: CU-mask micro (unbuffered), A/B of fuse_ref_last, refine parity
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5c
rm -rf "$O"; mkdir -p "$O"
cd "$R"
hipcc --offload-arch=gfx950 -O3 -o /tmp/cumask tools/micro/cumask.hip > "$O/cumask_build.log" 2>&1
timeout 300 /tmp/cumask > "$O/micro_cumask.txt" 2>&1; echo "cumask rc=$?" >> "$O/micro_cumask.txt"; cat "$O/micro_cumask.txt"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "refine or schedule_options or bitexact_vs_c_oracle or full_size" > "$O/pytest_refine.txt" 2>&1; tail -4 "$O/pytest_refine.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']; print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms ref_dws', k.get('ref_dws',{}).get('avg_us'), 'ref_last', k.get('ref_last',{}).get('avg_us'))"; }
for rep in 1 2 3; do
  for v in 1 0; do
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt fuse_ref_last=$v 2>/dev/null | line "B=1 fuse_ref_last=$v rep$rep" >> "$O/ab_fuse_ref_last.txt"
  done
done
for v in 1 0; do
  python bench.py --batch 2 --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt fuse_ref_last=$v 2>/dev/null | line "B=2 fuse_ref_last=$v" >> "$O/ab_fuse_ref_last.txt"
  python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt fuse_ref_last=$v 2>/dev/null | line "B=8 fuse_ref_last=$v" >> "$O/ab_fuse_ref_last.txt"
  python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --opt fuse_ref_last=$v 2>/dev/null | line "cfg3 fuse_ref_last=$v" >> "$O/ab_fuse_ref_last.txt"
done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined 2>/dev/null | line "B=1 driver flags (defaults)" >> "$O/ab_fuse_ref_last.txt"
cat "$O/ab_fuse_ref_last.txt"
