#!/bin/bash
# round 5, call p: the launch plan re-swept after this round's changes (every option against the default), B=1 / B=8 / config 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5p
rm -rf "$O"; mkdir -p "$O"
cd "$R"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"; }
run() {  # tag, bench args, options...
  tag=$1; shift; args=$1; shift
  o=""; for x in "$@"; do o="$o --opt $x"; done
  python bench.py $args --no-cpu-baseline --no-pipelined $o 2>/dev/null | line "$tag [$*]" >> "$O/plan_sweep_after_r05.txt"
}
for pass in 1 2; do
  for opts in "" "left_at=0" "defer_upsample=0" "fuse_first=0" "fuse_shift=0" "conv3d_order=0" "fuse_last1=0" "fuse_ref_last=0" "fork_ext=0" "fork2_after=0" "mid8_tile=3" "mid8_balance=0" "warp_form=0" "split_heads=1"; do
    run "B=1 pass$pass" "--steps 200 --warmup 10" $opts
  done
  for opts in "" "left_at=0" "split_heads=1" "ref_pipe=1" "ref_chunk_mb=36" "ref_chunk_mb=144" "fork2_after=0" "fork2_after=3" "tail_at=0" "fuse_ref_last=1" "mid8_balance=0" "conv3d_order=0"; do
    run "B=8 pass$pass" "--batch 8 --steps 40 --warmup 5" $opts
  done
done
for opts in "" "ref_pipe=0" "left_at=0" "ref_chunk_mb=144" "fork2_after=0" "fork2_after=3" "split_heads=1"; do
  run "cfg3" "--batch 8 --size 368x1232 --steps 10 --warmup 3" $opts
done
cat "$O/plan_sweep_after_r05.txt"
