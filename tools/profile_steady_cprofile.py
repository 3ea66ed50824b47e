#!/usr/bin/env python3
"""cProfile of the steady-state frame loop of bench.py --api reference (host side): where the ~1.2 ms of host time per frame go."""
import cProfile, pstats, os, sys, io, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devo_amd import synth, altcorr, fastba, projective_ops as pops
from devo_amd.lietorch import SE3
from devo_amd.update import Update
import devo_amd.backends as B
B.install()
dev = "cuda"
M, mem, H, W, C, nk = 96, 32, 120, 160, 128, 40
g_ii, g_jj, g_kk = [t.to(dev) for t in synth.sliding_window_graph(nk, M)]
E = g_ii.numel()
sposes = synth.make_poses(48, 1, trans_step=0.01, rot_step=0.002).to(dev)
spatches = synth.make_patches(48, M, H, W, seed=1)[0].to(dev)
intr = synth.make_intrinsics(48, H, W).to(dev)
dt = torch.float16
fmap1_ = (torch.randn(1, mem, C, H, W, device=dev) / 4).to(dt)
fmap2_ = (torch.randn(1, mem, C, H // 4, W // 4, device=dev) / 4).to(dt)
gmap_ = (torch.randn(mem, M, C, 3, 3, device=dev) / 4).to(dt)
gm = gmap_.view(1, mem * M, C, 3, 3)
imap_ = (torch.randn(mem, M, 384, device=dev) * 0.5).to(dt)
fm_new, f1_new, gm_new = fmap1_[:, 0].clone(), fmap2_[:, 0].clone(), gmap_[0].clone()
upd = Update(3).to(dev).eval()
with torch.no_grad():
    for p_ in upd.parameters():
        if p_.dim() == 2 and p_.shape[0] == 2: p_.mul_(0.05)
lm = torch.tensor([1e-4], device=dev)
st = {"f": nk, "net": torch.zeros(1, E, 384, device=dev, dtype=dt)}
P1, Q1 = sposes.clone(), spatches.clone()

def frame():
    ii, jj, kk = g_ii.clone(), g_jj.clone(), g_kk.clone()
    ring_idx = kk % (M * mem)
    k = st["f"] % mem; st["f"] += 1
    gmap_[k] = gm_new; fmap1_[:, k] = fm_new; fmap2_[:, k] = f1_new
    P1.copy_(sposes); Q1.copy_(spatches)
    coords = pops.transform(SE3(P1), Q1, intr, ii, jj, kk, fused=True).permute(0, 1, 4, 2, 3).contiguous()
    with torch.autocast("cuda", enabled=True, dtype=torch.float16):
        ii1, jj1 = ring_idx, jj % mem
        corr = torch.stack([altcorr.corr(gm, fmap1_, coords / 1, ii1, jj1, 3), altcorr.corr(gm, fmap2_, coords / 4, ii1, jj1, 3)], -1).view(1, E, -1)
        ctx = imap_.view(1, mem * M, 384)[:, ring_idx]
        st["net"], (delta, weight, _) = upd(st["net"], ctx, corr, None, ii, jj, kk)
    target = coords[..., 1, 1] + delta.float()
    fastba.BA(P1, Q1, intr, target, weight.float(), lm, ii, jj, kk, nk - 10, nk, 2)

with torch.no_grad():
    for _ in range(10): frame()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(30): frame()
    pr.disable()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"host {th / 30 * 1e3:.3f} ms per frame (under cProfile)")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:60]))
