"""Debug: phase times of k_ba_accumulate_reg (needs the trace build:  tools/build_variant.sh acctrace ba -DDEVO_ACC_TRACE,
then  DEVO_LIB=devo_amd/lib/libdevo_acctrace.so python tools/acc_trace.py).  Prints, over all waves of the last launch, when each
phase ended relative to the first wave's start (100 MHz stamps -> µs)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import devo_amd._lib as L
if os.environ.get("DEVO_LIB"): L.LIB_PATH = os.path.abspath(os.environ["DEVO_LIB"])
from devo_amd import synth
from devo_amd.backends import cuda_ba

dev = "cuda"

cfg = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
n, M, H, W = cfg["n"], cfg["M"], cfg["H"], cfg["W"]
poses, (patches, _), intr = synth.make_poses(n, 1234), synth.make_patches(n, M, H, W, seed=1234), synth.make_intrinsics(n, H, W)
ii, jj, kk = (t.to(dev) for t in synth.full_graph(n, M))
poses, patches, intr = poses.to(dev), patches.to(dev), intr.to(dev)
E, Np = ii.numel(), patches.shape[1]
coords = cuda_ba.transform(poses, patches, intr, ii, jj, kk)
g = torch.Generator().manual_seed(7)
target = coords[:, :, 1, 1, :] + torch.randn(1, E, 2, generator=g).to(dev)
weight = torch.rand(1, E, 2, generator=g).to(dev)
lmbda = torch.tensor([1e-4], device=dev)
ws = cuda_ba.workspace(E, Np, n - 1, dev)
for _ in range(3):
    p, q = poses.clone(), patches.clone()
    cuda_ba.forward(p, q, intr, target, weight, lmbda, ii, jj, kk, 1, n, 1, ws=ws)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8 * 16))()
lib = L.lib()
rc = lib.devo_debug_acc_trace(buf)
assert rc == 0, rc
t = np.array(buf, dtype=np.float64).reshape(256, 8, 16)
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
names = {0: "entry", 1: "LDS zeroed + barrier", 2: "edge slots loaded", 3: "edge terms, patch 1", 8: "  regular? + slots", 9: "  scratch + wave sums", 10: "  fold pass 1", 11: "  fold pass 2",
         4: "patch 1 folded (+ rhs)", 5: "edge terms, patch 2", 6: "patch 2 folded", 7: "partials written"}
for i, nm in names.items():
    v = t[:, :, i]
    v = (v[v > 0] - t0) / 100.0
    if v.size == 0: continue
    print(f"{nm:24s} waves {v.size:5d}  first {v.min():6.2f}  mean {v.mean():6.2f}  last {v.max():6.2f} us")
