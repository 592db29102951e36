"""Host cost of one forward (no device sync inside the loop) vs the device time per forward, by layer of the host stack:
model() (the Python shim: input checks, four output allocations, ctypes call), ops.forward, and the bare C call lws_forward
into preallocated outputs; the same with option side_streams = 0 (every launch on one stream: no event records / waits)."""
import ctypes, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lwsnet_amd                                   # noqa: F401  (exports the runtime switches before HIP initialises)
import torch
from lwsnet_amd import _lib, ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.synth import make_batch
from lwsnet_amd.weights import default_args, make_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
l, r = make_batch(B, 256, 512, 0)
l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
lib = _lib.load()
outs = [torch.empty((B, 1, 256, 512), device=dev) for _ in range(4)]
ptrs = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in outs])
lp, rp = ctypes.c_void_p(l.data_ptr()), ctypes.c_void_p(r.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def bare():
    rc = lib.lws_forward(m._h, lp, rp, B, 256, 512, ptrs, st)
    assert rc == 0


for side in (1, 0):
    m.set_option("side_streams", side)
    for _ in range(30):
        m(l, r)
    torch.cuda.synchronize()
    for name, fn in (("model()", lambda: m(l, r)), ("model(out=)", lambda: m(l, r, out=outs)), ("ops.forward", lambda: ops.forward(m._h, l, r, outs)),
                     ("lws_forward (C)", bare)):
        N = 400
        t0 = time.perf_counter()
        for _ in range(N):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"B={B} side_streams={side} {name:16s}: host {1e6 * (t1 - t0) / N:7.1f} us per call (loop returns), device-complete {1e6 * (t2 - t0) / N:7.1f} us per call")

# Without back-pressure: the loops above run the host ahead of the device until the hardware queue is full, after which every
# launch call blocks and "host" converges to the device time.  Here: K calls issued into an EMPTY queue (device idle), host only.
m.set_option("side_streams", 1)
for K in (1, 2, 4):
    ts = []
    for _ in range(40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            bare()
        ts.append(1e6 * (time.perf_counter() - t0) / K)
    ts.sort()
    print(f"B={B} lws_forward (C), {K} call(s) into an empty queue: host {ts[len(ts) // 2]:7.1f} us per call (median of 40; min {ts[0]:.1f})")
# ... and where inside the call the host time goes: the same with the side-stream work switched off piecewise
for opts, what in (({"side_streams": 0}, "one stream, no events"),):
    for k, v in opts.items():
        m.set_option(k, v)
    ts = []
    for _ in range(40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bare()
        ts.append(1e6 * (time.perf_counter() - t0))
    ts.sort()
    print(f"B={B} lws_forward (C), 1 call into an empty queue, {what}: host {ts[len(ts) // 2]:7.1f} us (min {ts[0]:.1f})")
    m.set_option("side_streams", 1)
