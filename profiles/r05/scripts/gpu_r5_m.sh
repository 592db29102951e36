#!/bin/bash
# round 5, call m: k_conv3d_mid8q with half tasks (one output-channel group per wave): alone and in the forward
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5m
rm -rf "$O"; mkdir -p "$O"
cd "$R"
for t in 0 5 6; do
  python tools/sbench.py --opt mid8_tile=$t 2>/dev/null | grep "mid8_form=1" >> "$O/sbench_mid8_half.txt"
  python tools/sbench.py --batch 2 --opt mid8_tile=$t 2>/dev/null | grep "mid8_form=1" >> "$O/sbench_mid8_half.txt"
done
python tools/sbench.py --batch 8 --opt mid8_tile=5 2>/dev/null | grep "mid8_form=1" >> "$O/sbench_mid8_half.txt"
python tools/sbench.py --batch 8 --opt mid8_tile=0 2>/dev/null | grep "mid8_form=1" >> "$O/sbench_mid8_half.txt"
cat "$O/sbench_mid8_half.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms mid8', d['secondary'] and (d['secondary']['stage2']['avg_launch_us'], d['secondary']['stage3']['avg_launch_us']))"; }
for rep in 1 2 3; do
  for t in 0 5 6 13 45; do    # 13 = stage 2 half 3x2 + stage 3 3x2 full (5 + 8*1); 45 = stage 2 half + stage 3 half (5 + 8*5)
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_tile=$t 2>/dev/null | line "B=1 mid8_tile=$t rep$rep" >> "$O/bench_mid8_half.txt"
  done
done
for t in 0 5; do python bench.py --batch 2 --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --opt mid8_tile=$t 2>/dev/null | line "B=2 mid8_tile=$t" >> "$O/bench_mid8_half.txt"; done
cat "$O/bench_mid8_half.txt"
