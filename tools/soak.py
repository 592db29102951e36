"""Soak: repeated forwards must be bit-identical (stream/event races would show up as run-to-run differences)."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device('cuda:0')
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
bad = 0
for B, H, W in [(1, 256, 512), (2, 64, 256), (3, 128, 256), (8, 256, 512), (1, 368, 1232)]:
    l, r = make_batch(B, H, W, 3)
    l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    ref = [p.clone() for p in m(l, r)]
    torch.cuda.synchronize()
    for it in range(200 if B == 1 else 40):
        out = m(l, r)
        if it % 7 == 0:
            torch.cuda.synchronize()
        for s in range(4):
            if not torch.equal(out[s], ref[s]):
                bad += 1
                print("MISMATCH", B, H, W, "iter", it, "stage", s + 1, float((out[s] - ref[s]).abs().max()))
    torch.cuda.synchronize()
    print("ok", B, H, W)
print("soak mismatches:", bad)
sys.exit(1 if bad else 0)
