#!/bin/bash
# round 5, call d: CU-mask micro with per-XCD budgets; side_cus sweep at B=8 / 8x368x1232 / B=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5d
rm -rf "$O"; mkdir -p "$O"
cd "$R"
hipcc --offload-arch=gfx950 -O3 -o /tmp/cumask tools/micro/cumask.hip > "$O/cumask_build.log" 2>&1
timeout 120 /tmp/cumask > "$O/micro_cumask.txt" 2>&1; echo "cumask rc=$?" >> "$O/micro_cumask.txt"; cat "$O/micro_cumask.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']; print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms mid16', d['roofline']['avg_launch_us'], 'mid8', d['secondary'] and (d['secondary']['stage2']['avg_launch_us'], d['secondary']['stage3']['avg_launch_us']), 'ref_dws', k.get('ref_dws',{}).get('avg_us'), 'conv64', k.get('ref_conv64',{}).get('avg_us'))"; }
for x in 0 4 8 12 16 24; do
  timeout 300 python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt side_cus=$x 2>/dev/null | line "B=8 side_cus=$x" >> "$O/sweep_side_cus.txt"
done
cat "$O/sweep_side_cus.txt"
for x in 0 8 16 24; do
  timeout 300 python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --opt side_cus=$x 2>/dev/null | line "cfg3 side_cus=$x" >> "$O/sweep_side_cus.txt"
  timeout 300 python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --opt side_cus=$x --opt ref_pipe=0 2>/dev/null | line "cfg3 side_cus=$x ref_pipe=0" >> "$O/sweep_side_cus.txt"
done
for x in 0 8 16; do
  timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt side_cus=$x 2>/dev/null | line "B=1 side_cus=$x" >> "$O/sweep_side_cus.txt"
done
for x in 0 8 16; do
  timeout 300 python bench.py --batch 4 --steps 50 --warmup 10 --no-cpu-baseline --no-pipelined --opt side_cus=$x 2>/dev/null | line "B=4 side_cus=$x" >> "$O/sweep_side_cus.txt"
done
cat "$O/sweep_side_cus.txt"
