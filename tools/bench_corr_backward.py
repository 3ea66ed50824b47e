#!/usr/bin/env python3
"""altcorr backward at the training configuration (BASELINE configuration 3: n = 15, M = 80, E = 18 000; gradients through 20 % of the
edges, correlation.py:20-25), per pyramid level.  Default: the product form (corr_bwd_mfma.h); DEVO_CORR_BWD_ATOMIC=1: the one-kernel
atomic path; DEVO_CORR_BWD_SEG=1: the segment-reduced path (per-edge kernel for d_fmap1 + LDS tile kernel for d_fmap2).   python tools/bench_corr_backward.py [keep fraction]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import devo_amd._lib as _L
if os.environ.get("DEVO_LIB"): _L.LIB_PATH = os.path.abspath(os.environ["DEVO_LIB"])      # A/B builds (tools/build_variant.sh)
from devo_amd import synth, altcorr
from devo_amd.backends import cuda_ba, cuda_corr

keep = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
cfg = synth.workload("cfg2_m80")
n, M, H, W, C, R = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["C"], cfg["R"]
dev = torch.device("cuda", 0)
poses = synth.make_poses(n, 1234).to(dev)
patches, centres = synth.make_patches(n, M, H, W, seed=1234)
intr = synth.make_intrinsics(n, H, W).to(dev)
ii, jj, kk = [t.to(dev) for t in synth.full_graph(n, M)]
fmap, gmap = synth.make_features(n, M, C, H, W, centres, seed=1234)
f0 = fmap.to(dev); f1 = synth.pyramid_l1(f0)
coords = cuda_ba.transform(poses, patches.to(dev), intr, ii, jj, kk, layout="2pp")                     # [1, E, 2, 3, 3]
E = ii.numel()
sel = torch.rand(E, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) < keep
c, k2, j2 = coords[:, sel].contiguous(), kk[sel], jj[sel]
g = torch.randn(1, int(sel.sum()), 7, 7, 3, 3, device=dev)
mode = "segment-reduced" if os.environ.get("DEVO_CORR_BWD_SEG") else "atomic" if os.environ.get("DEVO_CORR_BWD_ATOMIC") else "product form"
nchw = os.environ.get("DEVO_BWD_NCHW") == "1"       # the reference's own NCHW tensors (enet.py:203-216): cuda_corr.backward's cached channels-last copy
lay = (lambda t: t.contiguous()) if nchw else altcorr.channels_last
if nchw: mode += ", NCHW inputs"
for lvl, (fm, s) in enumerate(((lay(f0), 1.0), (lay(f1), 4.0))):
    cs = (c / s).contiguous()
    gm = gmap.to(dev)
    for _ in range(3): cuda_corr.backward(gm, fm, cs, k2, j2, g, R)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); ev0.record()
    for _ in range(20): cuda_corr.backward(gm, fm, cs, k2, j2, g, R)
    ev1.record(); torch.cuda.synchronize()
    print(f"{mode}: level {lvl}, {int(sel.sum())} of {E} edges: {ev0.elapsed_time(ev1) / 20 * 1e3:.1f} us per backward (incl. the memsets / scratch)", flush=True)
