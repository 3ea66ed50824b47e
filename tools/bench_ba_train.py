#!/usr/bin/env python3
"""Time ONE differentiable Gauss-Newton step (devo_amd.ba.BA, BASELINE configuration 3: n = 15, M = 80, E = 18 000, fp32):
forward alone and forward + backward, on the fused HIP solve (devo_ba_solve_terms / _backward) and on the torch composition
(DEVO_BA_TORCH=1), plus the share of the reprojection + Jacobians (projective_ops.transform(jacobian=True)) that both paths
run through autograd."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs
from devo_amd import synth, projective_ops as pops
from devo_amd.ba import BA
from devo_amd.lietorch import SE3

dev = torch.device("cuda", 0)
cfg = synth.workload("cfg2_m80")
d, _ = build_inputs(cfg, 1234, dev, torch.float32, "cl")
n, H, W = cfg["n"], cfg["H"], cfg["W"]
ii, jj, kk = d["ii"], d["jj"], d["kk"]
bounds = [-64, -64, W + 64, H + 64]
with torch.no_grad():
    coords = pops.transform(SE3(d["poses0"]), d["patches0"], d["intr"], ii, jj, kk)
target0 = coords[..., 1, 1, :] + d["delta"]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]


def timed(fn_fwd, backward, reps=20):
    for _ in range(3):
        out = fn_fwd()
        if backward:
            out.backward()
    torch.cuda.synchronize()
    tf = tb = 0.0
    for _ in range(reps):
        ev[0].record(); out = fn_fwd(); ev[1].record()
        if backward:
            out.backward()
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    return tf / reps, tb / reps


def ba_loss():
    tgt = target0.clone().requires_grad_(True)
    wgt = d["weight"].clone().requires_grad_(True)
    G, P = BA(SE3(d["poses0"].clone()), d["patches0"].clone(), d["intr"], tgt, wgt, 1e-4, ii, jj, kk, bounds, ep=10.0, fixedp=1, n_frames=n)
    return (G.data ** 2).sum() + (P[:, :, 2] ** 2).sum()


def jac_loss():
    P = d["patches0"].clone().requires_grad_(True)
    c, ok, (Ji, Jj, Jz) = pops.transform(SE3(d["poses0"].clone().requires_grad_(True)), P, d["intr"], ii, jj, kk, jacobian=True)
    return (c ** 2).sum() + (Ji ** 2).sum() + (Jj ** 2).sum() + (Jz ** 2).sum()


for name, env in (("fused HIP solve", "0"), ("torch composition", "1")):
    os.environ["DEVO_BA_TORCH"] = env
    with torch.no_grad():
        f0, _ = timed(ba_loss, False)
    f, b = timed(ba_loss, True)
    print(f"BA step, {name:18s}: forward (no grad) {f0:.3f} ms | with autograd: forward {f:.3f} ms, backward {b:.3f} ms, total {f + b:.3f} ms")
f, b = timed(jac_loss, True)
print(f"of which transform(jacobian=True) through autograd (inputs requiring grad): forward {f:.3f} ms, backward {b:.3f} ms")
