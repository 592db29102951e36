#!/bin/bash
# round 5, call i: runtime switches that touch fences / signals: do they change the cost of forks, live joins, a launch?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5i
rm -rf "$O"; mkdir -p "$O"
cd "$R"
hipcc --offload-arch=gfx950 -O3 -o /tmp/event_cost tools/micro/event_cost.hip
for e in "NONE=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" "AMD_OPT_FLUSH=0" "DEBUG_CLR_SKIP_RELEASE_SCOPE=1" "ROC_ACTIVE_WAIT_TIMEOUT=100" "GPU_STREAMOPS_CP_WAIT=1" "DEBUG_HIP_DYNAMIC_QUEUES=0" "ROC_CPU_WAIT_FOR_SIGNAL=0"; do
  echo "== $e" >> "$O/micro_event_cost_env.txt"
  env $e timeout 120 /tmp/event_cost 2>&1 | grep -E "base|record|fork|join|ext" >> "$O/micro_event_cost_env.txt"
done
cat "$O/micro_event_cost_env.txt"
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms mid16', d['roofline']['avg_launch_us'])"; }
for e in "NONE=1" "ROC_SYSTEM_SCOPE_SIGNAL=0" "AMD_OPT_FLUSH=0" "DEBUG_CLR_SKIP_RELEASE_SCOPE=1" "ROC_ACTIVE_WAIT_TIMEOUT=100"; do
  env $e timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined 2>/dev/null | line "B=1 $e" >> "$O/bench_env.txt"
done
cat "$O/bench_env.txt"
