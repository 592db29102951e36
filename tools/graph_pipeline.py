"""Experiment: S captured HIP graphs of LWSNet.forward (B=1), one per handle/stream, replayed round-robin --
the host cost of a forward drops from ~345 us of launches to one graph launch, so the S-way overlap is GPU-bound."""
import os, sys, time
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
ref = None
for S in (1, 2, 3, 4):
    models = [LWSNet(default_args(), device=dev).set_state_dict(sd).eval() for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    graphs, outs = [], []
    for m, st in zip(models, streams):
        lib.lws_reserve(m._h, 1, 256, 512)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(3):
                m(l, r)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            o = m(l, r)
        graphs.append(g)
        outs.append(o)
    torch.cuda.synchronize()
    for i in range(4 * S):
        with torch.cuda.stream(streams[i % S]):
            graphs[i % S].replay()
    torch.cuda.synchronize()
    if ref is None:
        ref = [p.clone() for p in models[0](l, r)]
        torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for o in outs for a, b in zip(o, ref))
    N = 300
    t0 = time.perf_counter()
    for i in range(N):
        with torch.cuda.stream(streams[i % S]):
            graphs[i % S].replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"S={S}: {N / dt:8.1f} pairs/s ({1e6 * dt / N:6.1f} us/step; host {1e6 * (t1 - t0) / N:5.1f} us per replay); outputs equal eager: {ok}")
