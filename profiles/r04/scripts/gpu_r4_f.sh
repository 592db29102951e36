#!/bin/bash
# Round 4, call F: k_conv3d_mid8q with small tiles + capped residency on small grids (option mid8_balance) -- parity, timing.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4f
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "conv3d_stack or schedule_options" > "$O/pytest.txt" 2>&1; tail -5 "$O/pytest.txt"
for f in 0 1; do
  for b in 1 2 4; do
    python tools/sbench.py --batch $b --opt mid8_balance=$f 2>/dev/null | sed "s/^/mid8_balance=$f B=$b: /" | grep "mid8_form=1"
  done
  for i in 1 2; do python bench.py --no-cpu-baseline --no-pipelined --steps 200 --opt mid8_balance=$f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=1 mid8_balance=$f', d['value'], d['ms_per_step'], d['secondary']['stage2']['avg_launch_us'], d['secondary']['stage3']['avg_launch_us'])"; done
  for b in 2 4; do python bench.py --no-cpu-baseline --no-pipelined --batch $b --steps 100 --opt mid8_balance=$f 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=$b mid8_balance=$f', d['value'], d['ms_per_step'], d['secondary']['stage2']['avg_launch_us'], d['secondary']['stage3']['avg_launch_us'])"; done
done 2>&1 | tee "$O/mid8_balance.txt"
python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=1 default incl. pool', d['value'], d['ms_per_step'], d['pipelined'])" | tee -a "$O/mid8_balance.txt"
python tools/stamps.py mid8q3 1 > "$O/stamps_mid8q3_b1_balanced.txt" 2>/dev/null; python -m lwsnet_amd.build --force > /dev/null 2>&1; cat "$O/stamps_mid8q3_b1_balanced.txt"
