#!/usr/bin/env python3
"""The Update operator's differentiable (torch-composition) path at the training size: forward and backward of one call at E = 18 000
edges, and the share of the hipBLASLt GEMMs in it.  python tools/bench_update_train.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import synth
from devo_amd.update import Update
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n, M = 15, 80
ii, jj, kk = [t.to(dev) for t in synth.full_graph(n, M)]
E = ii.numel()
up = Update(3).to(dev).train()
net = torch.randn(1, E, 384, device=dev, requires_grad=True)
inp = torch.randn(1, E, 384, device=dev) * 0.1
corr = torch.randn(1, E, 882, device=dev, requires_grad=True)
def fwd():
    out, (d, w, _) = up(net, inp, corr, None, ii, jj, kk)
    return out.square().mean() + d.square().mean() + w.mean()
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
with torch.no_grad():
    t_ng = timed(lambda: up.forward_torch(net, inp, corr, ii, jj, kk))
t_f = timed(fwd)
def fb():
    for q in up.parameters(): q.grad = None
    net.grad = None; corr.grad = None
    fwd().backward()
t_fb = timed(fb)
X = torch.randn(E, 384, device=dev); W = torch.randn(384, 384, device=dev)
t_mm = timed(lambda: X @ W, 50)
print(f"Update (torch composition) at E = {E}: forward without grad {t_ng:.2f} ms, forward with autograd {t_f:.2f} ms, forward + backward {t_fb:.2f} ms; "
      f"one 18000 x 384 x 384 fp32 GEMM {t_mm * 1e3:.0f} us (24 Linear layers: ~{24 * 3 * t_mm:.2f} ms of GEMMs per forward + backward)")
