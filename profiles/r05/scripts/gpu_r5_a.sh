#!/bin/bash
# round 5, call a: the GPU suite at HEAD (incl. the new two-ranks-on-one-GPU tests), default bench lines, and trace-only
# timelines of one forward at B=8 and at 8x368x1232
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
timeout 2700 python -m pytest tests -x -q -m gpu > "$O/pytest.txt" 2>&1; tail -5 "$O/pytest.txt"
python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_b1_driver_flags.json" 2> "$O/bench_b1.err"; tail -c 400 "$O/bench_b1_driver_flags.json"
python bench.py --gpus 2 --one-gpu --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > "$O/bench_two_ranks_one_gpu_b8.json" 2> "$O/bench_two_ranks.err"; tail -c 600 "$O/bench_two_ranks_one_gpu_b8.json"
python bench.py --batch 8 --no-cpu-baseline --no-pipelined --steps 30 > "$O/bench_b8.json" 2> /dev/null
python bench.py --batch 8 --size 368x1232 --no-cpu-baseline --no-pipelined --steps 10 --warmup 3 > "$O/bench_cfg3.json" 2> /dev/null
cd /tmp; export TMPDIR=/tmp
for cfg in "b8:--batch 8 --steps 12 --warmup 3:14" "cfg3:--batch 8 --size 368x1232 --steps 6 --warmup 2:8"; do
  tag=${cfg%%:*}; rest=${cfg#*:}; args=${rest%%:*}; back=${rest##*:}
  rocprofv3 --kernel-trace --output-format csv -d "$O/kt_$tag" -o run -- python3 "$R/bench.py" $args --no-cpu-baseline --no-pipelined --spinup 0.1 > /dev/null 2> "$O/kt_$tag.err"
  python3 "$R/tools/timeline.py" "$O/kt_$tag/run_kernel_trace.csv" $back > "$O/timeline_${tag}_trace_only.txt" 2>&1
  rm -rf "$O/kt_$tag"
done
cd "$R"
for f in bench_b8 bench_cfg3; do python -c "
import json
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['secondary'] and (d['secondary']['stage2']['frac'], d['secondary']['stage3']['frac']))"; done
du -sh "$O"
