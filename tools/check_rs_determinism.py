#!/usr/bin/env python3
"""The row-resident kernels of csrc/gemm_rs.hip have no atomics: the same inputs must give the same bits every time.  Repeats the fp16
Update operator (three chain kernels + tails) and the single layers on the same inputs and compares bit for bit — a difference would be
a missing barrier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import update as UA, synth
dev = torch.device("cuda", 0)
bad = 0
for n, M, seed in ((15, 96, 5), (7, 13, 6), (32, 64, 7)):
    torch.manual_seed(seed)
    ii, jj, kk = [t.to(dev) for t in synth.full_graph(n, M)]
    E = ii.numel()
    upd = UA.Update(3).to(dev).half().eval()
    net = torch.randn(1, E, 384, device=dev).half() * 0.5; inp = torch.randn(1, E, 384, device=dev).half() * 0.5; corr = torch.randn(1, E, 882, device=dev).half()
    ref = None
    for rep in range(30):
        with torch.no_grad():
            o, (d, w, _) = upd(net, inp, corr, None, ii, jj, kk)
        cur = (o.clone(), d.clone(), w.clone())
        if ref is None:
            ref = cur
        elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
            bad += 1
            print(f"E {E}: repetition {rep} differs: max |diff| {max((a.float() - b.float()).abs().max().item() for a, b in zip(ref, cur)):.3e}", flush=True)
    x32 = torch.randn(E, 384, device=dev); lin = torch.nn.Linear(384, 384).to(dev)
    r0 = None
    for rep in range(20):
        with torch.no_grad():
            y = UA._linear_split(x32, lin.weight, lin.bias, rs=True).clone()
        if r0 is None: r0 = y
        elif not torch.equal(r0, y):
            bad += 1; print(f"E {E}: fp32 row-resident layer differs at repetition {rep}", flush=True)
    print(f"E {E}: done", flush=True)
print("differences:", bad)
sys.exit(1 if bad else 0)
