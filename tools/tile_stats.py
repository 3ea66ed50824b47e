#!/usr/bin/env python3
"""CPU simulation of the tile lookup's grouping: union regions of G consecutive plan edges, M-tiles per edge, bytes staged.
python tools/tile_stats.py [--workload cfg2] [--group 32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from devo_amd import synth
from oracle import pops
from oracle.lie import SE3

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2"); ap.add_argument("--group", type=int, default=32)
ap.add_argument("--band", type=int, default=16); ap.add_argument("--xw", type=int, default=8)
ap.add_argument("--maxpos", type=int, default=1024); ap.add_argument("--morton", type=int, default=0); ap.add_argument("--tmax", type=int, default=8)
args = ap.parse_args()
cfg = synth.workload(args.workload)
n, M, H, W, R = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["R"]
poses = synth.make_poses(n, 1234)
patches, _ = synth.make_patches(n, M, H, W, seed=1234)
intr = synth.make_intrinsics(n, H, W)
ii, jj, kk = synth.full_graph(n, M)
with torch.no_grad():
    coords = pops.transform(SE3(poses.double()), patches.double(), intr.double(), ii, jj, kk)   # [1,E,3,3,2]
coords = coords[0].reshape(-1, 9, 2).numpy()
E = coords.shape[0]
D = 2 * R + 2
cx, cy = coords[:, 4, 0], coords[:, 4, 1]
def boxes(s):
    c = np.floor(coords / s).astype(np.int64)
    x0, x1 = c[:, :, 0].min(1) - R, c[:, :, 0].max(1) - R + D
    y0, y1 = c[:, :, 1].min(1) - R, c[:, :, 1].max(1) - R + D
    return x0, y0, x1, y1
b0 = boxes(1.0); b1 = boxes(4.0)
for lvl, b in ((0, b0), (1, b1)):
    w, h = b[2] - b[0], b[3] - b[1]
    pos = w * h
    t = (pos + 15) // 16
    print(f"level {lvl}: box {w.mean():.2f} x {h.mean():.2f}, positions {pos.mean():.1f}, M-tiles mean {t.mean():.2f}; share by tiles: "
          + ", ".join(f"{k}:{(t == k).mean() * 100:.1f}%" for k in range(1, 16) if (t == k).any()) + f"; > {args.tmax}: {(t > args.tmax).mean() * 100:.2f}%")
# plan order: (frame, block of 4 bands x 64 px, band, column bin) of the patch centre at level 0
band = np.clip(np.clip(cy, 0, H - 1).astype(np.int64) // args.band, 0, None)
xb = np.clip(cx, 0, 1e6).astype(np.int64) // args.xw
bx = max(64 // args.xw, 1)
key = ((jj.numpy() * 1000 + (band // 4) * 40 + xb // bx) * 4 + band % 4) * bx + xb % bx
if args.morton:
    def spread(v):
        r = np.zeros_like(v)
        for i in range(10): r |= ((v >> i) & 1) << (2 * i)
        return r
    key = jj.numpy() * (1 << 22) + (spread(band) << 1 | spread(xb))
t0 = (b0[2] - b0[0]) * (b0[3] - b0[1]); t1 = (b1[2] - b1[0]) * (b1[3] - b1[1])
heavy = ((t0 + 15) // 16 > args.tmax) | ((t1 + 15) // 16 > args.tmax)
inimg = (b0[0] >= -64) & (b0[1] >= -64) & (b0[2] <= W + 64) & (b0[3] <= H + 64)
order = np.argsort(key[~heavy], kind="stable")
idx = np.nonzero(~heavy)[0][order]
print(f"edges {E}, heavy {heavy.sum()} ({heavy.mean() * 100:.2f}%)")
G = args.group
tot = {0: 0, 1: 0}; over = 0; rounds = 0; ngroups = 0
sizes0, sizes1 = [], []
for g in range(0, len(idx), G):
    sel = idx[g:g + G]
    ngroups += 1
    # greedy rounds: consecutive edges of one frame whose union regions fit maxpos at level 0 (clipped to the image)
    s = 0
    while s < len(sel):
        e = s
        fr = jj[sel[s]].item()
        X0 = Y0 = 10**9; X1 = Y1 = -10**9; U0 = V0 = 10**9; U1 = V1 = -10**9
        while e < len(sel) and jj[sel[e]].item() == fr:
            k = sel[e]
            nX0, nY0, nX1, nY1 = min(X0, b0[0][k]), min(Y0, b0[1][k]), max(X1, b0[2][k]), max(Y1, b0[3][k])
            nU0, nV0, nU1, nV1 = min(U0, b1[0][k]), min(V0, b1[1][k]), max(U1, b1[2][k]), max(V1, b1[3][k])
            if (nX1 - nX0) * (nY1 - nY0) > args.maxpos or (nU1 - nU0) * (nV1 - nV0) > args.maxpos:
                if e == s: raise SystemExit("single edge does not fit")
                break
            X0, Y0, X1, Y1, U0, V0, U1, V1 = nX0, nY0, nX1, nY1, nU0, nV0, nU1, nV1
            e += 1
        rounds += 1
        cw = max(min(X1, W) - max(X0, 0), 0); ch = max(min(Y1, H) - max(Y0, 0), 0)
        dw = max(min(U1, W // 4) - max(U0, 0), 0); dh = max(min(V1, H // 4) - max(V0, 0), 0)
        tot[0] += cw * ch; tot[1] += dw * dh
        sizes0.append((X1 - X0) * (Y1 - Y0)); sizes1.append((U1 - U0) * (V1 - V0))
        s = e
frame0, frame1 = n * H * W, n * (H // 4) * (W // 4)
print(f"group {G}: groups {ngroups}, rounds {rounds} ({rounds / ngroups:.2f} per group); region positions L0 mean {np.mean(sizes0):.0f} max {np.max(sizes0)}, "
      f"L1 mean {np.mean(sizes1):.0f} max {np.max(sizes1)}")
print(f"  staged positions (in-image) L0 {tot[0]} = {tot[0] / frame0:.2f} x the pyramid level, L1 {tot[1]} = {tot[1] / frame1:.2f} x; "
      f"bytes at 128 ch fp32: {(tot[0] + tot[1]) * 512 / 1e6:.0f} MB (per-edge boxes: {(t0.sum() + t1.sum()) * 512 / 1e6:.0f} MB)")
