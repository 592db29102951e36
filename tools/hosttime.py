"""Host cost of one model(left, right) call (no device sync inside the loop) vs the device time per forward."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lwsnet_amd import ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device("cuda:0")
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
l, r = make_batch(1, 256, 512, 0)
l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
for _ in range(20):
    m(l, r)
torch.cuda.synchronize()
for name, fn in (("model()", lambda: m(l, r)), ("ops.forward", lambda: ops.forward(m._h, l, r))):
    N = 300
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:12s}: host {1e6 * (t1 - t0) / N:7.1f} us per call (loop returns), device-complete {1e6 * (t2 - t0) / N:7.1f} us per call")
