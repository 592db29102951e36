#!/bin/bash
# round 5, call b: CU-mask map (tools/micro/cumask), parity of the stage-1 fused last layer / two-level deferred map,
# A/B of fuse_last1 at B=1, side_xcds sweep at B=8 and 8x368x1232
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5b
rm -rf "$O"; mkdir -p "$O"
cd "$R"
hipcc --offload-arch=gfx950 -O3 -o /tmp/cumask tools/micro/cumask.hip > "$O/cumask_build.log" 2>&1 && timeout 120 /tmp/cumask > "$O/micro_cumask.txt" 2>&1; cat "$O/micro_cumask.txt"
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$O/pytest_parity.txt" 2>&1; tail -5 "$O/pytest_parity.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms mid16', d['roofline']['avg_launch_us'], 'us', 'sec', d['secondary'] and (d['secondary']['stage2']['avg_launch_us'], d['secondary']['stage3']['avg_launch_us']))"; }
for rep in 1 2 3; do
  for v in 1 0; do
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt fuse_last1=$v 2>/dev/null | line "B=1 fuse_last1=$v rep$rep" >> "$O/ab_fuse_last1.txt"
  done
done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined 2>/dev/null | line "B=1 driver flags" >> "$O/ab_fuse_last1.txt"
cat "$O/ab_fuse_last1.txt"
for rep in 1 2; do
  for x in 0 1 2 3 4 6; do
    python bench.py --batch 8 --steps 30 --warmup 5 --no-cpu-baseline --no-pipelined --opt side_xcds=$x 2>/dev/null | line "B=8 side_xcds=$x rep$rep" >> "$O/sweep_side_xcds.txt"
  done
done
for x in 0 2 4 6; do
  python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --opt side_xcds=$x 2>/dev/null | line "cfg3 side_xcds=$x" >> "$O/sweep_side_xcds.txt"
done
for x in 0 2 4; do
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined --opt side_xcds=$x 2>/dev/null | line "B=1 side_xcds=$x" >> "$O/sweep_side_xcds.txt"
done
cat "$O/sweep_side_xcds.txt"
du -sh "$O"
