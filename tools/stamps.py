"""Diagnostic: per-workgroup phase stamps (s_memtime, shader cycles) of one kernel.

    python tools/stamps.py <kernel> [batch]     # rebuilds the library with -DLWS_STAMPS=<id>, runs, prints medians
kernels: mid16 mid16x mid8q2 mid8q3 mid8x2 mid8x3 conv64x last1 last3 first1 first3 dws conv64 feat pair0..pair3 ref_last warp2 warp3
Slots: 0..3 = phase boundaries as placed in the kernel source (LWS_STAMPK); 5 = hardware id of the CU (HW_ID | XCC_ID << 32);
6, 7 = s_memrealtime at the first / latest stamp.
CAUTION: a stamped build is a different kernel.  The inline asm can change its register allocation (k_ref_last: 129 registers in
the shipped build, 300 with stamps -- one workgroup per CU instead of three): compare `.amdhsa_next_free_vgpr` of the two builds
(hipcc --save-temps) before reading residency or phase times off a stamped run."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNELS = {  # name: (stamp id, translation unit, driver, arg)
    "mid16": (1, "conv3d", "stack", 0),
    "last1": (3, "conv3d", "stack", 0), "last3": (3, "conv3d", "stack", 2),
    "first1": (4, "conv3d", "stack", 0), "first3": (4, "conv3d", "stack", 2),
    
    "dws": (5, "conv2d", "refine", None), "conv64": (6, "conv2d", "refine", None),
    "feat": (7, "conv2d", "feat", None), "pair0": (13, "conv2d", "feat", None), "pair1": (14, "conv2d", "feat", None),
    "pair2": (15, "conv2d", "feat", None), "pair3": (16, "conv2d", "feat", None), "ref_last": (9, "conv2d", "refine", None),
    "warp2": (10, "volume", "stages", None), "warp3": (10, "volume", "stages", None),
    "mid8q3": (18, "conv3d", "stack", 2), "mid8q2": (18, "conv3d", "stack", 1),
    "mid16x": (19, "conv3d", "stack", 0), "mid8x3": (20, "conv3d", "stack", 2), "mid8x2": (20, "conv3d", "stack", 1),
    "conv64x": (21, "conv2d", "refine", None),
}
what = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kid, TU, driver, arg = KERNELS[what]
env = dict(os.environ, LWS_EXTRA_FLAGS=f"-DLWS_STAMPS={kid}")
subprocess.check_call([sys.executable, "-m", "lwsnet_amd.build", "--force"], env=env, stdout=subprocess.DEVNULL, cwd=ROOT)
import numpy as np, torch
from lwsnet_amd import _lib, ops
from lwsnet_amd.models import LWSNet
from lwsnet_amd.weights import default_args, make_state_dict
dev = torch.device('cuda:0')
m = LWSNet(default_args(), device=dev).set_state_dict(make_state_dict(7)).eval()
lib = ctypes.CDLL(_lib.LIB_PATH)
if kid in (19, 20, 21):
    m.set_option("split_bf16", {19: 1, 20: 2, 21: 4}[kid])
if driver == "stack":
    shape = [(B, 24, 32, 64), (B, 9, 64, 128), (B, 9, 128, 256)][arg]
    c = torch.rand(shape, device=dev) * 12
    for _ in range(5):
        ops.conv3d_stack(m._h, arg, c)
elif driver == "refine":
    left = torch.randn((B, 3, 256, 512), device=dev)
    p3 = torch.rand((B, 1, 256, 512), device=dev) * 100
    for _ in range(3):
        ops.refine(m._h, left, p3)
elif driver == "feat":
    img = torch.randn((2 * B, 3, 256, 512), device=dev)
    for _ in range(3):
        ops.feature_extraction(m._h, img)
elif driver == "stages":
    rng = np.random.default_rng(0)
    shapes = [(B, 16, 32, 64), (B, 16, 64, 128), (B, 8, 128, 256)]
    fl = [torch.from_numpy(np.abs(rng.standard_normal(s)).astype(np.float32)).to(dev) for s in shapes]
    fr = [torch.from_numpy(np.abs(rng.standard_normal(s)).astype(np.float32)).to(dev) for s in shapes]
    for _ in range(3):
        ops.disparity_stages(m._h, fl, fr, 256, 512)
torch.cuda.synchronize()
n = 4096 * 8
buf = (ctypes.c_ulonglong * n)()
assert getattr(lib, "lws_debug_read_stamps_" + TU)(buf, n) == 0
s = np.array(buf, dtype=np.int64).reshape(-1, 8)
s = s[(s[:, 0] > 0) & (s[:, 3] > 0)]
last = 3            # (slot 5 holds the hardware id of the CU since round 3; kernels stamp phases 0..3)
hw = s[:, 5]
cu = ((hw >> 32) & 15) * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 12) & 1) * 8 + ((hw >> 8) & 15)      # xcc, se, sh, cu

print(f"{what} B={B}: workgroups with stamps: {len(s)} (last launch of this kernel)")
rt = (s[:, 7] - s[:, 6]).astype(np.float64)                       # 100 MHz ticks between the first and the last stamp of a workgroup
ok = rt > 50
clk = (s[ok, last] - s[ok, 0]) / (rt[ok] * 10.0)                  # shader cycles per ns = GHz
print(f"  in-kernel clock (d s_memtime / d s_memrealtime, per workgroup): median {np.median(clk):.3f} GHz  p10 {np.percentile(clk, 10):.3f}  p90 {np.percentile(clk, 90):.3f}")
# s_memrealtime is one chip-wide clock: the launch's wall time and the number of workgroups in flight follow from it
t0, t1 = s[:, 6].min(), s[:, 7].max()
print(f"  launch (first workgroup start -> last stamped workgroup end): {(t1 - t0) / 100.0:.1f} us; sum of workgroup lifetimes / that = {rt.sum() / max(t1 - t0, 1):.1f} workgroups in flight on average")
if (hw > 0).any():
    ids, counts = np.unique(cu, return_counts=True)
    # peak number of this kernel's workgroups alive at once on one CU
    peak = 0
    for c in ids:
        ev = sorted([(a, 1) for a in s[cu == c, 6]] + [(b, -1) for b in s[cu == c, 7]])
        n = 0
        for _, d in ev:
            n += d
            peak = max(peak, n)
    if os.environ.get("LWS_STAMPS_VERBOSE"):
        for c in ids[:6]:
            iv = sorted(((a - t0) / 100.0, (b - t0) / 100.0) for a, b in zip(s[cu == c, 6], s[cu == c, 7]))
            print(f"    CU {c}: " + "  ".join(f"[{a:.1f}, {b:.1f}]" for a, b in iv))
    print(f"  CUs used: {len(ids)}; workgroups per CU: min {counts.min()} median {int(np.median(counts))} max {counts.max()}; most workgroups alive at once on one CU: {peak}")
st = (s[:, 6] - t0) / 100.0
print("  workgroup start times after the first (us): " + "  ".join(f"p{q}={np.percentile(st, q):.1f}" for q in (10, 25, 50, 75, 90, 100)))
for a, b in [(i, i + 1) for i in range(last)] + [(0, last)]:
    d = s[:, b] - s[:, a]
    print(f"  stamp {a}->{b}: median {np.median(d):8.0f}  p10 {np.percentile(d,10):8.0f}  p90 {np.percentile(d,90):8.0f} cycles")
