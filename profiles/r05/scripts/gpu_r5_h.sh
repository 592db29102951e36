#!/bin/bash
# round 5, call h: fork_ext / tail_at A/B at B=1 (and B=2, 4, 8), schedule-option parity
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5h
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule_options or bitexact_vs_c_oracle or batch_paths or repeatable or pool or profiler" > "$O/pytest_sched.txt" 2>&1; tail -4 "$O/pytest_sched.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms mid16', d['roofline']['avg_launch_us'])"; }
for rep in 1 2 3; do
  for o in "fork_ext=1 tail_at=0" "fork_ext=0 tail_at=0" "fork_ext=1 tail_at=1" "fork_ext=0 tail_at=1"; do
    set -- $o
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt $1 --opt $2 2>/dev/null | line "B=1 $o rep$rep" >> "$O/ab_fork_tail.txt"
  done
done
for B in 2 4 8; do
  for o in "fork_ext=1 tail_at=0" "fork_ext=1 tail_at=1" "fork_ext=0 tail_at=1"; do
    set -- $o
    python bench.py --batch $B --steps 60 --warmup 10 --no-cpu-baseline --no-pipelined --opt $1 --opt $2 2>/dev/null | line "B=$B $o" >> "$O/ab_fork_tail.txt"
  done
done
cat "$O/ab_fork_tail.txt"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('driver flags', d['value'], d['ms_per_step'], 'pipelined', d['pipelined']['value'])"
