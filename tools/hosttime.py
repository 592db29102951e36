import sys, time, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from lwsnet_amd import _lib, ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device('cuda:0')
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
l, r = make_batch(1, 256, 512, 0)
l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
for _ in range(10): m(l, r)
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N): m(l, r)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e6*(t1-t0)/N:.1f} us/step, total {1e6*(t2-t0)/N:.1f} us/step")
# raw C call only
lib = _lib.load()
preds = [torch.empty((1,1,256,512), device=dev) for _ in range(4)]
arr = (ctypes.c_void_p*4)(*[p.data_ptr() for p in preds])
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
t0 = time.perf_counter()
for _ in range(N): lib.lws_forward(m._h, ctypes.c_void_p(l.data_ptr()), ctypes.c_void_p(r.data_ptr()), 1, 256, 512, arr, st)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"C-only: issue {1e6*(t1-t0)/N:.1f} us/step, total {1e6*(t2-t0)/N:.1f} us/step")
