#!/usr/bin/env python3
"""cProfile of the host side of the reference's call sequence (bench.py:reference_api_probe, fp16, package form): where the ~260 us per
update iteration go (the sequence is host-bound)."""
import cProfile
import os
import pstats
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devo_amd import synth, altcorr, fastba, projective_ops as pops
from devo_amd.lietorch import SE3

dev = torch.device("cuda", 0)
cfg = synth.workload("cfg2")
n, M, H, W, C = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["C"]
mem, dt = 32, torch.float16
poses = synth.make_poses(n, 1234)
patches, centres = synth.make_patches(n, M, H, W, seed=1234)
intr = synth.make_intrinsics(n, H, W).to(dev)
ii, jj, kk = [t.to(dev) for t in synth.full_graph(n, M)]
fmap, gmap = synth.make_features(n, M, C, H, W, centres, seed=1234)
delta, weight = [t.to(dev) for t in synth.make_update_outputs(len(ii), 1234)]
lmbda = torch.as_tensor([1e-4], device=dev)
E = ii.numel()
fmap1_ = torch.zeros(1, mem, C, H, W, dtype=dt, device=dev)
fmap2_ = torch.zeros(1, mem, C, H // 4, W // 4, dtype=dt, device=dev)
gmap_ = torch.zeros(mem, M, C, 3, 3, dtype=dt, device=dev)
f0 = fmap.to(dev)
fmap1_[:, :n] = f0.to(dt); fmap2_[:, :n] = synth.pyramid_l1(f0).to(dt)
gmap_.view(1, mem * M, C, 3, 3)[:, :n * M] = gmap.to(dev).to(dt)
pyramid, gm = (fmap1_, fmap2_), gmap_.view(1, mem * M, C, 3, 3)
P0, Q0 = poses.to(dev), patches.to(dev)
P, Q = P0.clone(), Q0.clone()


def update():
    P.copy_(P0); Q.copy_(Q0)
    coords = pops.transform(SE3(P), Q, intr, ii, jj, kk)
    coords = coords.permute(0, 1, 4, 2, 3).contiguous()
    ii1 = kk % (M * mem); jj1 = jj % mem
    corr1 = altcorr.corr(gm, pyramid[0], coords / 1, ii1, jj1, 3)
    corr2 = altcorr.corr(gm, pyramid[1], coords / 4, ii1, jj1, 3)
    corr = torch.stack([corr1, corr2], -1).view(1, E, -1)
    target = coords[..., 1, 1] + delta.float()
    fastba.BA(P, Q, intr, target, weight, lmbda, ii, jj, kk, 1, n, 2)
    return corr


with torch.no_grad():
    for _ in range(10):
        update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        update()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"host {th / 200 * 1e6:.1f} us per iteration, wall {(time.perf_counter() - t0) / 200 * 1e6:.1f}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        update()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
