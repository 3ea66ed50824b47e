#!/usr/bin/env python3
"""Union-box statistics of the lookup at a bench workload (GPU): how many 64-position passes the matrix-core kernel
runs per edge and level, and how full they are.  python tools/box_stats.py [--workload cfg2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import synth
from devo_amd.backends import cuda_ba

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="cfg2"); args = ap.parse_args()
cfg = synth.workload(args.workload)
n, M, H, W, R = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["R"]
dev = torch.device("cuda")
poses = synth.make_poses(n, 1234).to(dev)
patches, _ = synth.make_patches(n, M, H, W, seed=1234)
intr = synth.make_intrinsics(n, H, W).to(dev)
ii, jj, kk = [t.to(dev) for t in synth.full_graph(n, M)]
coords = cuda_ba.transform(poses, patches.to(dev), intr, ii, jj, kk, layout="2pp")[0]              # feature-map pixels, [E,2,3,3]
D = 2 * R + 2
for lvl, s in ((0, 1.0), (1, 4.0)):
    c = torch.floor(coords / s)
    x, y = c[:, 0].reshape(-1, 9), c[:, 1].reshape(-1, 9)
    w = (x.max(1).values - x.min(1).values + D).long()
    h = (y.max(1).values - y.min(1).values + D).long()
    pos = w * h
    passes = (pos + 63) // 64
    print(f"level {lvl}: box {w.float().mean():.2f} x {h.float().mean():.2f}, positions mean {pos.float().mean():.1f}, "
          f"passes mean {passes.float().mean():.3f}; share of edges by passes: "
          + ", ".join(f"{p}: {(passes == p).float().mean() * 100:.1f}%" for p in range(1, 6))
          + f"; slot fill {pos.sum().item() / (64 * passes.sum().item()) * 100:.1f}%; needed 9*64 per edge vs computed 12*64*passes: "
          f"{9 * 64 / (12 * 64 * passes.float().mean().item()) * 100:.1f}%")
