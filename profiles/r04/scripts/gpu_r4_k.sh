#!/bin/bash
# Round 4, call K: the launch-plan options re-swept at B=8 / config 3 after this round's kernel changes (all bit-identical).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4k
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1
one() {  # label, bench args...
  python bench.py --no-cpu-baseline --no-pipelined "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
one "B=8 default" --batch 8 --steps 30
one "B=8 left_at=0" --batch 8 --steps 30 --opt left_at=0
one "B=8 split_heads=1" --batch 8 --steps 30 --opt split_heads=1
one "B=8 ref_pipe=1" --batch 8 --steps 30 --opt ref_pipe=1
one "B=8 ref_chunk_mb=36" --batch 8 --steps 30 --opt ref_chunk_mb=36
one "B=8 ref_chunk_mb=36 ref_pipe=1" --batch 8 --steps 30 --opt ref_chunk_mb=36 --opt ref_pipe=1
one "B=8 ref_chunk_mb=144" --batch 8 --steps 30 --opt ref_chunk_mb=144
one "B=8 defer_upsample=0/side_streams=0" --batch 8 --steps 30 --opt side_streams=0
done 2>&1 | tee "$O/plan_sweep_b8.txt"
for rep in 1 2; do
one "cfg3 default" --batch 8 --size 368x1232 --steps 10 --warmup 3
one "cfg3 ref_pipe=0" --batch 8 --size 368x1232 --steps 10 --warmup 3 --opt ref_pipe=0
one "cfg3 left_at=0" --batch 8 --size 368x1232 --steps 10 --warmup 3 --opt left_at=0
one "cfg3 split_heads=1" --batch 8 --size 368x1232 --steps 10 --warmup 3 --opt split_heads=1
one "cfg3 ref_chunk_mb=144" --batch 8 --size 368x1232 --steps 10 --warmup 3 --opt ref_chunk_mb=144
one "cfg3 ref_chunk_mb=250" --batch 8 --size 368x1232 --steps 10 --warmup 3 --opt ref_chunk_mb=250
done 2>&1 | tee "$O/plan_sweep_cfg3.txt"
for rep in 1 2; do
one "B=1 default" --steps 200
one "B=1 left_at=0" --steps 200 --opt left_at=0
one "B=1 defer_upsample=0" --steps 200 --opt defer_upsample=0
one "B=1 fuse_first=0" --steps 200 --opt fuse_first=0
one "B=1 conv3d_order=0" --steps 200 --opt conv3d_order=0
done 2>&1 | tee "$O/plan_sweep_b1.txt"
