"""Soak: thousands of forwards issued back to back into rotating output buffers (batches 1, 2, 3, 8; three geometries), every 50th
step all buffers compared bit for bit with the first result -- a race between the caller's stream and the side stream (forks bound to
kernel completion signals, deferred maps written by their consumers) would show up as a sporadic mismatch.  python tools/soak.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lwsnet_amd, torch, numpy as np
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device("cuda:0")
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
bad = 0
for B, H, W, n in ((1, 256, 512, 3000), (2, 256, 512, 800), (1, 64, 256, 3000), (3, 128, 384, 500), (8, 256, 512, 150)):
    l, r = make_batch(B, H, W, 11)
    l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    ref = [p.clone() for p in m(l, r)]
    outs = [[torch.empty_like(ref[0]) for _ in range(4)] for _ in range(3)]
    for i in range(n):
        o = outs[i % 3]
        m(l, r, out=o)
        if i % 50 == 49:
            torch.cuda.synchronize()
            for k in range(3):
                for s in range(4):
                    if not torch.equal(outs[k][s], ref[s]):
                        bad += 1
    torch.cuda.synchronize()
    print(f"B={B} {H}x{W}: {n} forwards back to back, mismatching maps: {bad}")
print("SOAK", "OK" if bad == 0 else "FAILED")
