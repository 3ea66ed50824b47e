#!/usr/bin/env python3
"""CPU simulation on cfg2's coordinates: how many 16-position tiles two / three plan neighbours would need if they shared one union box per
level (the verdict's form (b), not built: DESIGN.md 3.1f).  python tools/pair_stats.py"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from devo_amd import synth
from oracle import pops
from oracle.lie import SE3
cfg = synth.workload("cfg2")
n, M, H, W, R = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["R"]
poses = synth.make_poses(n, 1234); patches, _ = synth.make_patches(n, M, H, W, seed=1234); intr = synth.make_intrinsics(n, H, W)
ii, jj, kk = synth.full_graph(n, M)
c = pops.transform(SE3(poses.double()), patches.double(), intr.double(), ii, jj, kk)   # [1,E,3,3,2]
c = c[0].reshape(-1, 9, 2).float()
D = 2 * R + 2
def boxes(s):
    f = torch.floor(c / s)
    x0 = f[..., 0].min(1).values - R; x1 = f[..., 0].max(1).values - R + D
    y0 = f[..., 1].min(1).values - R; y1 = f[..., 1].max(1).values - R + D
    return x0, x1, y0, y1
Hs, Ws = {0: (H, W), 1: (H // 4, W // 4)}, None
for name, keyf in (("REAL plan (frame, 16-row band, 8-px column bin, then edge order)", lambda cx, cy: (cy // 16) * 100000 + (cx // 8) * 8),
                   ("(frame, 8-row band, 8-px column bin, then edge order)", lambda cx, cy: (cy // 8) * 100000 + (cx // 8) * 8),
                   ("(frame, 8-row band, 4-px column bin, then edge order)", lambda cx, cy: (cy // 8) * 100000 + (cx // 4) * 4),
                   ("(frame, 4-row band, 4-px column bin, then edge order)", lambda cx, cy: (cy // 4) * 100000 + (cx // 4) * 4),
                   ("plan (frame, 16-row band, x)", lambda cx, cy: (cy // 16) * 100000 + cx),
                   ("(frame, 8-row band, x)", lambda cx, cy: (cy // 8) * 100000 + cx),
                   ("(frame, 4-row band, x)", lambda cx, cy: (cy // 4) * 100000 + cx),
                   ("(frame, 6-row band, x)", lambda cx, cy: (cy // 6) * 100000 + cx)):
    cx, cy = c[:, 4, 0].clamp(0, W - 1), c[:, 4, 1].clamp(0, H - 1)
    key = jj.double() * 1e9 + keyf(cx.double().floor(), cy.double().floor())
    order = torch.argsort(key, stable=True)
    tot = {}
    for grp in (2, 3):
        o = order[: (len(order) // grp) * grp].reshape(-1, grp)
        samef = (jj[o] == jj[o][:, :1]).all(1)
        for lvl, s in ((0, 1.0), (1, 4.0)):
            x0, x1, y0, y1 = boxes(s)
            single = ((x1 - x0) * (y1 - y0))
            ux0, ux1 = x0[o].min(1).values, x1[o].max(1).values
            uy0, uy1 = y0[o].min(1).values, y1[o].max(1).values
            upos = (ux1 - ux0) * (uy1 - uy0)
            spos = single[o].sum(1)
            cap = 160 if grp == 2 else 256
            ok = samef & (upos <= cap)
            stiles = ((single[o] + 15) // 16).sum(1)
            utiles = (upos + 15) // 16
            # cost in tiles if eligible groups are shared, others processed singly
            cost = torch.where(ok, utiles, stiles).sum().item()
            print(f"{name} groups of {grp}, level {lvl}: eligible {ok.float().mean()*100:.1f}%, tiles {cost / stiles.sum().item():.3f} of single "
                  f"(union positions / sum {upos[ok].sum().item() / spos[ok].sum().item():.3f} on eligible)")
