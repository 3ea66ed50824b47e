#!/usr/bin/env python3
"""CPU simulation of corr_fwd_region_kernel's grouping (corr_region.h): plan chunks -> local sort -> greedy rounds.
Reports rounds per chunk, edges per round, staged positions / bytes.  python tools/region_stats.py [--workload cfg2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from devo_amd import synth
from oracle import pops
from oracle.lie import SE3

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--chunk", type=int, default=24); ap.add_argument("--round", type=int, default=12)
ap.add_argument("--cap", type=int, default=512); ap.add_argument("--tmax", type=int, default=10)
ap.add_argument("--band", type=int, default=16); ap.add_argument("--xw", type=int, default=8); ap.add_argument("--blocks", type=int, default=1)
ap.add_argument("--sortband", type=int, default=16)
args = ap.parse_args()
cfg = synth.workload(args.workload)
n, M, H, W, R = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["R"]
poses = synth.make_poses(n, 1234)
patches, _ = synth.make_patches(n, M, H, W, seed=1234)
intr = synth.make_intrinsics(n, H, W)
ii, jj, kk = synth.full_graph(n, M)
with torch.no_grad():
    coords = pops.transform(SE3(poses.double()), patches.double(), intr.double(), ii, jj, kk)
coords = coords[0].reshape(-1, 9, 2).numpy(); jjn = jj.numpy()
E = coords.shape[0]; D = 2 * R + 2
def boxes(s):
    c = np.floor(coords / s).astype(np.int64)
    return c[:, :, 0].min(1) - R, c[:, :, 1].min(1) - R, c[:, :, 0].max(1) - R + D, c[:, :, 1].max(1) - R + D
b0 = boxes(1.0); b1 = boxes(4.0)
nt0 = ((b0[2] - b0[0]) * (b0[3] - b0[1]) + 15) // 16; nt1 = ((b1[2] - b1[0]) * (b1[3] - b1[1]) + 15) // 16
heavy = (nt0 > args.tmax) | (nt1 > args.tmax)
def clip(b, w, h):
    return np.clip(b[0], 0, w), np.clip(b[1], 0, h), np.clip(b[2], 0, w), np.clip(b[3], 0, h)
c0 = clip(b0, W, H); c1 = clip(b1, W // 4, H // 4)
live0 = (c0[2] > c0[0]) & (c0[3] > c0[1]); live1 = (c1[2] > c1[0]) & (c1[3] > c1[1])
dead = ~live0 & ~live1
heavy = heavy & ~dead
cx, cy = coords[:, 4, 0], coords[:, 4, 1]
band = np.clip(cy, 0, H - 1).astype(np.int64) // args.band
xb = np.clip(cx, 0, 1e6).astype(np.int64) // args.xw
bx = max(64 // args.xw, 1) if args.blocks else 10**6
BB = 4 if args.blocks else 1
key = (((jjn * 1000 + (band // BB)) * 1000 + xb // bx) * BB + band % BB) * 1000 + xb % bx
idx = np.nonzero(~heavy & ~dead)[0]; idx = idx[np.argsort(key[idx], kind="stable")]
L = len(idx); nchunks = max(1, int(round(L / args.chunk)))
rounds = 0; pos0 = pos1 = 0; hist = np.zeros(args.round + 1, int); chunk_rounds = []
for c in range(nchunks):
    sel = idx[c * L // nchunks:(c + 1) * L // nchunks]
    k2 = (jjn[sel] * 4096 + band[sel]) * 8192 + (b0[0][sel] + 2048)
    sel = sel[np.argsort(k2, kind="stable")]
    s = 0; nr = 0
    while s < len(sel):
        X = [10**9, 10**9, -10**9, -10**9]; U = [10**9, 10**9, -10**9, -10**9]; e = s
        while e < len(sel) and e - s < args.round and jjn[sel[e]] == jjn[sel[s]]:
            k = sel[e]
            nX = [min(X[0], c0[0][k]), min(X[1], c0[1][k]), max(X[2], c0[2][k]), max(X[3], c0[3][k])] if live0[k] else X
            nU = [min(U[0], c1[0][k]), min(U[1], c1[1][k]), max(U[2], c1[2][k]), max(U[3], c1[3][k])] if live1[k] else U
            if max(nX[2] - nX[0], 0) * max(nX[3] - nX[1], 0) > args.cap or max(nU[2] - nU[0], 0) * max(nU[3] - nU[1], 0) > args.cap: break
            X, U = nX, nU; e += 1
        hist[e - s] += 1; nr += 1
        pos0 += max(X[2] - X[0], 0) * max(X[3] - X[1], 0); pos1 += max(U[2] - U[0], 0) * max(U[3] - U[1], 0); s = e
    rounds += nr; chunk_rounds.append(nr)
f0, f1 = n * H * W, n * (H // 4) * (W // 4)
print(f"{args.workload}: edges {E}, dead {dead.sum()} ({dead.mean()*100:.1f}%), heavy {heavy.sum()} ({heavy.mean() * 100:.2f}%), chunks {nchunks} of ~{L / nchunks:.1f}, rounds {rounds} ({rounds / nchunks:.2f}/chunk, max {max(chunk_rounds)}), "
      f"edges/round {L / rounds:.2f}")
print("  rounds by edge count: " + ", ".join(f"{i}:{hist[i]}" for i in range(1, args.round + 1) if hist[i]))
print(f"  staged positions L0 {pos0} ({pos0 / f0:.2f}x level, {pos0 / rounds:.0f}/round), L1 {pos1} ({pos1 / f1:.2f}x, {pos1 / rounds:.0f}/round); "
      f"bytes (128 ch fp32) {(pos0 + pos1) * 512 / 1e6:.0f} MB vs per-edge {(((b0[2]-b0[0])*(b0[3]-b0[1])).sum() + ((b1[2]-b1[0])*(b1[3]-b1[1])).sum()) * 512 / 1e6:.0f} MB")
if os.environ.get("DUMP"):
    for c in (100, 101):
        sel = idx[c * L // nchunks:(c + 1) * L // nchunks]
        print("chunk", c, [(int(jjn[k]), int(b0[0][k]), int(b0[1][k]), int(b0[2][k]-b0[0][k]), int(b0[3][k]-b0[1][k]), int(cx[k]), int(cy[k])) for k in sel])
out_all = (b0[2] <= 0) | (b0[3] <= 0) | (b0[0] >= W) | (b0[1] >= H)
cen_out = (cx < 0) | (cy < 0) | (cx >= W) | (cy >= H)
print(f"  boxes entirely outside the frame: {out_all.mean() * 100:.1f}%; centre outside: {cen_out.mean() * 100:.1f}%; far outside (> 64 px): {((cx < -64) | (cy < -64) | (cx > W + 64) | (cy > H + 64)).mean() * 100:.1f}%")
