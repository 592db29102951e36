#!/bin/bash
# round 5, call f: k_conv3d_mid8q 3x2x32 tiles per stage and per batch, in the forward
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5f
rm -rf "$O"; mkdir -p "$O"
cd "$R"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms mid8', d['secondary'] and (d['secondary']['stage2']['avg_launch_us'], d['secondary']['stage3']['avg_launch_us']))"; }
for rep in 1 2; do
  for t in 0 1 8 24 9 25; do   # 8 = stage 3 only 3x2 (stage 2 auto); 1 = both; 24 = stage 3 3x4 forced, stage 2 auto; 9 = both 3x2; 25 = stage2 3x2 + stage3 3x4
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_tile=$t 2>/dev/null | line "B=1 mid8_tile=$t rep$rep" >> "$O/bench_mid8_tile_per_stage.txt"
  done
done
for B in 2 4 8; do
  for t in 0 1; do
    python bench.py --batch $B --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_tile=$t 2>/dev/null | line "B=$B mid8_tile=$t" >> "$O/bench_mid8_tile_per_stage.txt"
  done
done
for t in 0 1; do
  python bench.py --size 544x960 --maxdisp0 32 --feature-fp16 --no-cpu-baseline --no-pipelined --steps 50 --opt mid8_tile=$t 2>/dev/null | line "cfg5 mid8_tile=$t" >> "$O/bench_mid8_tile_per_stage.txt"
  python bench.py --size 368x1232 --no-cpu-baseline --no-pipelined --steps 50 --opt mid8_tile=$t 2>/dev/null | line "1x368x1232 mid8_tile=$t" >> "$O/bench_mid8_tile_per_stage.txt"
done
cat "$O/bench_mid8_tile_per_stage.txt"
