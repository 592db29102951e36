"""Experiment: throughput of B=1 forwards when consecutive steps run on S independent handles/streams."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device('cuda:0')
sd = make_state_dict(7)
l, r = make_batch(1, 256, 512, 0)
l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
for S in (1, 2, 3, 4):
    models = [LWSNet(default_args(), device=dev).set_state_dict(sd).eval() for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for i in range(10 * S):
        with torch.cuda.stream(streams[i % S]):
            models[i % S](l, r)
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for i in range(N):
        with torch.cuda.stream(streams[i % S]):
            models[i % S](l, r)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"S={S}: {N/dt:.1f} pairs/s ({1e6*dt/N:.1f} us/step)")

# in-situ k_conv3d_mid16 duration under S-way overlap
import ctypes
from lwsnet_amd import _lib
lib = _lib.load()
for S in (1, 2, 3):
    models = [LWSNet(default_args(), device=dev).set_state_dict(sd).eval() for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for i in range(10 * S):
        with torch.cuda.stream(streams[i % S]):
            models[i % S](l, r)
    torch.cuda.synchronize()
    for m in models:
        lib.lws_profile_enable(m._h, 1 << 3)
    N = 90
    t0 = time.perf_counter()
    for i in range(N):
        with torch.cuda.stream(streams[i % S]):
            models[i % S](l, r)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot_ms, n = 0.0, 0
    for m in models:
        tot = (ctypes.c_double * _lib.LWS_KC_COUNT)(); cnt = (ctypes.c_int64 * _lib.LWS_KC_COUNT)()
        lib.lws_profile_read(m._h, tot, cnt); lib.lws_profile_enable(m._h, 0)
        tot_ms += tot[3]; n += cnt[3]
    avg = 1e3 * tot_ms / n
    print(f"S={S}: {N/dt:.1f} pairs/s, mid16 avg {avg:.1f} us = {2.718e9/(avg*1e-6)/1e12:.1f} TF")
