"""Experiment: capture LWSNet.forward (B=1) into a HIP graph via torch.cuda.CUDAGraph and compare replay vs eager."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from lwsnet_amd import _lib
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device('cuda:0')
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
lib = _lib.load()
l, r = make_batch(1, 256, 512, 0)
l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
lib.lws_reserve(m._h, 1, 256, 512)
for _ in range(5):
    ref = m(l, r)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    out = m(l, r)
torch.cuda.synchronize()
print(f"eager : {1e6*(time.perf_counter()-t0)/N:.1f} us/step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m(l, r)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    gout = m(l, r)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph output equals eager:", all(torch.equal(a, b) for a, b in zip(gout, ref)))
t0 = time.perf_counter()
for _ in range(N):
    g.replay()
torch.cuda.synchronize()
print(f"graph : {1e6*(time.perf_counter()-t0)/N:.1f} us/step")
