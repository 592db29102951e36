#!/bin/bash
# round 6, call i: stage 1's fused last Conv3D layer + soft-argmin (24 x 2 x 4 tiles) at batches > 2 too, followed by
# k_upsample_add, against k_conv3d_last + k_softargmin_upsample (experiment value "fuse_last1" = 2), two passes; bit check.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6i
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -c "import __graft_entry__ as g; g.build(); print('build ok')" > "$O/build.log" 2>&1; tail -1 "$O/build.log"
python - <<PY
import torch, numpy as np
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device("cuda:0")
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
for B, H, W in ((3, 64, 256), (4, 256, 512), (2, 64, 256)):
    l, r = make_batch(B, H, W, 9)
    a = [p.clone() for p in m(l, r)]
    m.set_option("fuse_last1", 2)
    b = m(l, r)
    m.set_option("fuse_last1", 1)
    print(B, H, W, "bit-equal:", all(torch.equal(x, y) for x, y in zip(a, b)))
PY
run() {
  python bench.py --no-cpu-baseline --no-pipelined --no-measure-traffic $2 > "$O/$1.json" 2> "$O/$1.err"
  python -c "
import json
try:
    d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); k=d['kernels']
    print('$1', d['value'], d['ms_per_step'], 'last', k['conv3d_last']['avg_us'], 'softargmin', (k.get('softargmin') or {}).get('avg_us'), 'upsample', (k.get('upsample_add') or {}), 'clk', d['roofline']['clock_ghz'])
except Exception as e: print('$1 ERR', e, open('$O/$1.err').read()[-300:])"
}
for pass in 1 2 3; do
  for v in 1 2; do
    run "p${pass}_b8_fl$v" "--batch 8 --steps 40 --opt fuse_last1=$v"
    run "p${pass}_b4_fl$v" "--batch 4 --steps 60 --opt fuse_last1=$v"
    run "p${pass}_cfg3_fl$v" "--batch 8 --size 368x1232 --steps 12 --warmup 3 --opt fuse_last1=$v"
  done
done 2>&1 | tee "$O/ab_fuse_last1_large_batches.txt"
