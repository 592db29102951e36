#!/bin/bash
# Round 4, call E: k_ref_dws2 back (two refinement blocks per launch, batches <= 2) -- parity and what it is worth.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "refine or schedule_options or forward_bitexact or forward_batch or config1" > "$O/pytest.txt" 2>&1; tail -5 "$O/pytest.txt"
for f in 0 1; do
  python tools/rbench.py --opt fuse_dws=$f > "$O/rbench_b1_fuse$f.txt" 2>/dev/null; head -6 "$O/rbench_b1_fuse$f.txt"
  for i in 1 2; do python bench.py --no-cpu-baseline --no-pipelined --steps 200 --opt fuse_dws=$f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=1 fuse_dws=$f', d['value'], d['ms_per_step'], d['kernels'].get('ref_dws'))"; done
  python bench.py --no-cpu-baseline --no-pipelined --batch 2 --steps 100 --opt fuse_dws=$f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=2 fuse_dws=$f', d['value'], d['ms_per_step'], d['kernels'].get('ref_dws'))"
  python bench.py --no-cpu-baseline --no-pipelined --batch 4 --steps 60 --opt fuse_dws=$f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=4 fuse_dws=$f', d['value'], d['ms_per_step'], d['kernels'].get('ref_dws'))"
done 2>&1 | tee "$O/fuse_dws_bench.txt"
python bench.py --size 368x1232 --no-cpu-baseline --no-pipelined --steps 60 --opt fuse_dws=0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=1 368x1232 fuse_dws=0', d['value'], d['ms_per_step'])" | tee -a "$O/fuse_dws_bench.txt"
python bench.py --size 368x1232 --no-cpu-baseline --no-pipelined --steps 60 --opt fuse_dws=1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=1 368x1232 fuse_dws=1', d['value'], d['ms_per_step'])" | tee -a "$O/fuse_dws_bench.txt"
