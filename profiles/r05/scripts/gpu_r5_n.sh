#!/bin/bash
# round 5, call n: position of the second fork (behind stage 1's last layer / its 4th / 3rd middle layer), every batch shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5n
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule_options or forward_bitexact or batch_paths or repeatable" > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms mid16', d['roofline']['avg_launch_us'])"; }
for rep in 1 2 3; do
  for v in 0 4 3; do
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt fork2_after=$v 2>/dev/null | line "B=1 fork2_after=$v rep$rep" >> "$O/ab_fork2_after.txt"
  done
done
for rep in 1 2; do
  for B in 2 4 8; do
    for v in 0 4 3; do
      python bench.py --batch $B --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined --opt fork2_after=$v 2>/dev/null | line "B=$B fork2_after=$v rep$rep" >> "$O/ab_fork2_after.txt"
    done
  done
  for v in 0 4 3; do
    python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --opt fork2_after=$v 2>/dev/null | line "cfg3 fork2_after=$v rep$rep" >> "$O/ab_fork2_after.txt"
  done
done
for v in 0 4; do python bench.py --size 544x960 --maxdisp0 32 --feature-fp16 --steps 50 --no-cpu-baseline --no-pipelined --opt fork2_after=$v 2>/dev/null | line "cfg5 fork2_after=$v" >> "$O/ab_fork2_after.txt"; done
cat "$O/ab_fork2_after.txt"
