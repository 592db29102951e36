#!/bin/bash
# Round 4, call I: k_conv3d_mid8v (packed-float32 VALU form of the 8 -> 8 layers) -- parity, per-launch times against k_conv3d_mid8q.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4i
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "conv3d_stack or schedule_options" > "$O/pytest.txt" 2>&1; tail -5 "$O/pytest.txt"
for cfg in "--batch 1" "--batch 2" "--batch 8" "--batch 8 --size 368x1232" "--batch 1 --size 368x1232"; do
  python tools/sbench.py $cfg 2>/dev/null | grep -E "mid8_form=(1|3)"
done | tee "$O/sbench_mid8v.txt"
for f in 1 3; do
  python bench.py --no-cpu-baseline --no-pipelined --steps 200 --opt mid8_form=$f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=1 mid8_form=$f', d['value'], d['ms_per_step'], d['kernels']['conv3d_mid8'])"
  python bench.py --no-cpu-baseline --no-pipelined --batch 8 --steps 30 --opt mid8_form=$f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=8 mid8_form=$f', d['value'], d['ms_per_step'], d['kernels']['conv3d_mid8'])"
  python bench.py --no-cpu-baseline --no-pipelined --batch 8 --size 368x1232 --steps 10 --warmup 3 --opt mid8_form=$f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg3 mid8_form=$f', d['value'], d['ms_per_step'], d['kernels']['conv3d_mid8'])"
done | tee "$O/bench_mid8v.txt"
