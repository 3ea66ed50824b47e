#!/usr/bin/env python3
"""Host time (enqueue, no synchronisation) and total time of the pieces of one steady-state DEVO frame on the 45 312-edge sliding-window graph with NEW
index tensors per frame: the Update operator's graph tables, its forward under autocast, the BA, the lookups."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devo_amd import synth, altcorr, fastba, projective_ops as pops
from devo_amd.lietorch import SE3
from devo_amd.update import Update
from devo_amd.backends import cuda_ba
import devo_amd.backends as B
B.install()
dev = "cuda"
M, mem, H, W, C = 96, 32, 120, 160, 128
ii, jj, kk = [t.to(dev) for t in synth.sliding_window_graph(40, M)]
E = ii.numel()
poses = synth.make_poses(48, 1, trans_step=0.01, rot_step=0.002).to(dev)
patches = synth.make_patches(48, M, H, W, seed=1)[0].to(dev)
intr = synth.make_intrinsics(48, H, W).to(dev)
dt = torch.float16
fmap1_ = torch.randn(1, mem, C, H, W, device=dev).to(dt) / 4
fmap2_ = torch.randn(1, mem, C, H // 4, W // 4, device=dev).to(dt) / 4
gm = (torch.randn(1, mem * M, C, 3, 3, device=dev) / 4).to(dt)
imap = (torch.randn(1, mem * M, 384, device=dev) * 0.5).to(dt)
upd = Update(3).to(dev).eval()
net = torch.zeros(1, E, 384, device=dev, dtype=dt)
lm = torch.tensor([1e-4], device=dev)
delta, weight = [t.to(dev) for t in synth.make_update_outputs(E, 1, sigma=0.3)]

def timeit(name, fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print(f"{name:58s} host {th / n * 1e6:8.1f} us   total {ta / n * 1e6:8.1f} us")

with torch.no_grad():
    coords = pops.transform(SE3(poses), patches, intr, ii, jj, kk, fused=True).permute(0, 1, 4, 2, 3).contiguous()
    corr = torch.stack([altcorr.corr(gm, fmap1_, coords / 1, kk % (M * mem), jj % mem, 3), altcorr.corr(gm, fmap2_, coords / 4, kk % (M * mem), jj % mem, 3)], -1).view(1, E, -1)
    ctx = imap[:, kk % (M * mem)]
    timeit("clone ii / jj / kk", lambda: (ii.clone(), jj.clone(), kk.clone()))
    timeit("Update._tables on new index tensors", lambda: upd._tables(ii.clone(), jj.clone(), kk.clone()))
    timeit("  cuda_ba.neighbors", lambda: cuda_ba.neighbors(kk, jj))
    from devo_amd.update import _Groups
    timeit("  _Groups(kk)", lambda: _Groups(kk.long().contiguous()))
    def upd_same():
        with torch.autocast("cuda", dtype=torch.float16):
            upd(net, ctx, corr, None, ii, jj, kk)
    timeit("Update forward under autocast, SAME graph (tables cached)", upd_same)
    def upd_new():
        with torch.autocast("cuda", dtype=torch.float16):
            upd(net, ctx, corr, None, ii.clone(), jj.clone(), kk.clone())
    timeit("Update forward under autocast, new graph", upd_new)
    P, Q = poses.clone(), patches.clone()
    tgt = coords[:, :, :, 1, 1] + delta
    timeit("fastba.BA, same kk", lambda: fastba.BA(P, Q, intr, tgt, weight, lm, ii, jj, kk, 30, 40, 2))
    timeit("fastba.BA, new kk", lambda: fastba.BA(P, Q, intr, tgt, weight, lm, ii, jj, kk.clone(), 30, 40, 2))
    timeit("transform + permute", lambda: pops.transform(SE3(poses), patches, intr, ii, jj, kk, fused=True).permute(0, 1, 4, 2, 3).contiguous())
    def look():
        ii1, jj1 = kk % (M * mem), jj % mem
        return torch.stack([altcorr.corr(gm, fmap1_, coords / 1, ii1, jj1, 3), altcorr.corr(gm, fmap2_, coords / 4, ii1, jj1, 3)], -1).view(1, E, -1)
    timeit("two lookups + stack (DEVO.corr)", look)
    timeit("ctx = imap[:, kk % (M mem)]", lambda: imap[:, kk % (M * mem)])
