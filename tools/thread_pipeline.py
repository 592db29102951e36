"""Experiment: S host threads, each driving its own handle / HIP stream (ctypes releases the GIL inside lws_forward), so
the ~345 us of host launch work per forward runs S-way parallel and the overlap of consecutive B=1 forwards is GPU-bound."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lwsnet_amd import _lib
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device("cuda:0")
sd = make_state_dict(7)
lib = _lib.load()
l, r = make_batch(1, 256, 512, 0)
l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
for S in (1, 2, 3, 4, 6):
    models = [LWSNet(default_args(), device=dev).set_state_dict(sd).eval() for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    N = 200
    outs = [None] * S

    def work(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                outs[i] = models[i](l, r)

    ths = [threading.Thread(target=work, args=(i, 10)) for i in range(S)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    ths = [threading.Thread(target=work, args=(i, N)) for i in range(S)]
    t0 = time.perf_counter()
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(a, b) for o in outs for a, b in zip(o, outs[0]))
    print(f"S={S} threads: {S * N / dt:8.1f} pairs/s ({1e6 * dt / (S * N):6.1f} us/step); outputs equal: {same}")
