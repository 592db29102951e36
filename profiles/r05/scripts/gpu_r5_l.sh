#!/bin/bash
# round 5, call l: k_ref_dws with the row-walk depthwise (18 instead of 36 LDS reads per thread): parity, per-launch times, steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5l
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "refine or forward_bitexact or batch_paths or full_size or config3" > "$O/pytest_refine.txt" 2>&1; tail -3 "$O/pytest_refine.txt"
python tools/rbench.py 2>/dev/null | grep -E "ref_dws|ref_conv64|ref_last|bitwise" > "$O/rbench_b1.txt"
python tools/rbench.py --batch 8 --iters 20 2>/dev/null | grep -E "ref_dws|bitwise" > "$O/rbench_b8.txt"
python tools/rbench.py --batch 8 --size 368x1232 --iters 6 2>/dev/null | grep -E "ref_dws|bitwise" > "$O/rbench_b8_368x1232.txt"
cat "$O/rbench_b1.txt" "$O/rbench_b8.txt" "$O/rbench_b8_368x1232.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']; print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms ref_dws', k['ref_dws']['avg_us'])"; }
for r in 1 2 3; do python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined 2>/dev/null | line "B=1 rep$r" >> "$O/bench.txt"; done
for r in 1 2; do python bench.py --batch 8 --steps 40 --warmup 5 --no-cpu-baseline --no-pipelined 2>/dev/null | line "B=8 rep$r" >> "$O/bench.txt"; done
python bench.py --batch 8 --size 368x1232 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined 2>/dev/null | line "cfg3" >> "$O/bench.txt"
cat "$O/bench.txt"
