#!/usr/bin/env python3
"""Time one training-style update iteration (BASELINE.json config 3: batch 1, n = 15 frames, M = 80 patches/frame, fp32):
reprojection with autograd -> 2-level correlation lookup (channels-last pyramid, differentiable, 20 % edge dropout in the
backward) -> a stand-in for the Update MLP (two small linear maps so that gradients flow to corr / target / weight) ->
2 x differentiable BA (devo_amd.ba.BA over the HIP SE3 ops) -> pose + reprojection loss -> backward.  Forward and
backward times are reported separately.  Not the headline metric; evidence for SURVEY 8d's configuration 3."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs
from devo_amd import synth, altcorr, projective_ops as pops
from devo_amd.ba import BA
from devo_amd.lietorch import SE3

dev = torch.device("cuda", 0)
cfg = synth.workload("cfg2_m80")
d, _ = build_inputs(cfg, 1234, dev, torch.float32, "cl")
n, H, W, R = cfg["n"], cfg["H"], cfg["W"], cfg["R"]
ii, jj, kk = d["ii"], d["jj"], d["kk"]
E = ii.numel()
torch.manual_seed(0)
head = torch.nn.Linear(2 * 49 * 9, 4).to(dev)                      # stand-in for the Update MLP: corr -> (delta, weight logits)
fmaps = [f.clone().requires_grad_(True) for f in d["pyramid"]]
gmap = d["gmap"].clone().requires_grad_(True)
bounds = [-64, -64, W + 64, H + 64]


def forward():
    G = SE3(d["poses0"].clone())
    P = d["patches0"].clone()
    coords = pops.transform(G, P, d["intr"], ii, jj, kk)
    c2 = coords.permute(0, 1, 4, 2, 3).contiguous()
    corr = torch.stack([altcorr.corr(gmap, fmaps[0], c2 / 1, kk, jj, R, 0.2), altcorr.corr(gmap, fmaps[1], c2 / 4, kk, jj, R, 0.2)], -1).view(1, E, -1)
    o = head(corr)
    target = coords[..., 1, 1, :].detach() + 0.5 * torch.tanh(o[..., :2])
    weight = torch.sigmoid(o[..., 2:])
    for _ in range(2):
        G, P = BA(G, P, d["intr"], target, weight, 1e-4, ii, jj, kk, bounds, ep=10.0, fixedp=1)
    cf = pops.transform(G, P, d["intr"], ii, jj, kk)
    return (cf - coords.detach()).abs().mean() + (G.log() ** 2).mean()


for _ in range(2):
    forward().backward()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
reps = 5
for _ in range(reps):
    ev[0].record(); loss = forward(); ev[1].record(); loss.backward(); ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print(f"training-style iteration  E={E} (n={n}, M={cfg['M']}) fp32: forward {tf / reps:.2f} ms, backward {tb / reps:.2f} ms")
