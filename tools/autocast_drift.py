#!/usr/bin/env python3
"""how far does the Update operator's autocast shortcut (devo.py:311's call on an fp16-storage copy of the operator: the recurrent
`net` rounded to fp16 at every call and between the chain kernels) drift from what torch's own autocast composes (fp16 Linear layers, fp32 LayerNorms /
sums, `net` carried in fp32), over a long recurrence?  Both against the fp32 operator, `net` fed back for 64 iterations on fixed random inputs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devo_amd import synth
from devo_amd.update import Update
dev = "cuda"
ii, jj, kk = [t.to(dev) for t in synth.full_graph(8, 48)]
E = ii.numel()
torch.manual_seed(3)
upd = Update(3).to(dev).eval()
g = torch.Generator().manual_seed(4)
inp = (torch.randn(1, E, 384, generator=g) * 0.5).to(dev)
corrs = [(torch.randn(1, E, 882, generator=g) * 0.5).to(dev) for _ in range(4)]

def run(mode, iters=64):
    net = torch.zeros(1, E, 384, device=dev)
    outs = []
    with torch.no_grad():
        for it in range(iters):
            c = corrs[it % 4]
            if mode == "fp32":
                net, (d, w, _) = upd(net, inp, c, None, ii, jj, kk)
            elif mode == "shortcut":
                with torch.autocast("cuda", dtype=torch.float16):
                    net, (d, w, _) = upd(net, inp, c, None, ii, jj, kk)
            else:                                      # torch's autocast over the torch composition (what the reference's modules do)
                with torch.autocast("cuda", dtype=torch.float16):
                    net, (d, w, _) = upd.forward_torch(net, inp, c, ii, jj, kk)
            net = net.float()
            if it in (0, 3, 15, 63): outs.append((it + 1, net.clone(), d.float().clone(), w.float().clone()))
    return outs
ref = run("fp32")
for mode in ("shortcut", "torch-autocast"):
    print(mode)
    for (it, n, d, w), (_, n0, d0, w0) in zip(run(mode), ref):
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-6))
        print(f"  after {it:2d} iterations: net {rel(n, n0):.2e}   delta {rel(d, d0):.2e}   weight {float((w - w0).abs().max()):.2e}")
