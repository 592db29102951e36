#!/bin/bash
# round 5, call e: k_conv3d_mid8q tile shapes at B=1 alone (sbench) and in the forward
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5e
rm -rf "$O"; mkdir -p "$O"
cd "$R"
for t in 0 1 2 3 4; do
  python tools/sbench.py --opt mid8_tile=$t 2>/dev/null | grep "mid8_form=1" >> "$O/sbench_mid8_tile_b1.txt"
done
for t in 0 1 2; do
  python tools/sbench.py --batch 2 --opt mid8_tile=$t 2>/dev/null | grep "mid8_form=1" >> "$O/sbench_mid8_tile_b1.txt"
done
cat "$O/sbench_mid8_tile_b1.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms mid8', d['secondary'] and (d['secondary']['stage2']['avg_launch_us'], d['secondary']['stage3']['avg_launch_us']))"; }
for rep in 1 2; do
  for t in 0 1 2 3; do
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_tile=$t 2>/dev/null | line "B=1 mid8_tile=$t rep$rep" >> "$O/bench_mid8_tile.txt"
  done
done
cat "$O/bench_mid8_tile.txt"
