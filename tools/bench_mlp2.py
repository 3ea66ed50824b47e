#!/usr/bin/env python3
"""csrc/mlp2.hip (Linear - ReLU - Linear in one launch, fp16) against two launches of the fp16 Linear kernel (+ the gather kernel): HIP-graph
replays, us per chain at the update operator's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import update as UA, _lib as L
dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 21600


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for K1, gather in ((384, False), (384, True), (882, False)):
    l1, l2 = torch.nn.Linear(K1, 384).to(dev).half(), torch.nn.Linear(384, 384).to(dev).half()
    x = (torch.randn(rows, K1, device=dev) * 0.5).half()
    idx = torch.randint(0, rows, (rows,), device=dev) if gather else None
    res = torch.randn(rows, 384, device=dev).half() if gather else None

    def fused():
        return UA._mlp2_f16(x, l1, l2, residual=res, gather=idx)

    def two():
        t = x
        if gather:
            t = torch.empty_like(x)
            L.check(L.lib().devo_upd_masked_gather(L.ptr(x), L.ptr(idx), L.ptr(t), rows, K1, L.dtype_code(x), L.stream()), "g")
        h = UA.Update._lin(t, l1.weight, l1.bias, relu=True)
        if res is not None:
            r2 = res.clone()
            return UA.Update._lin(h, l2.weight, l2.bias, residual=r2)
        return UA.Update._lin(h, l2.weight, l2.bias)
    print(f"rows {rows} K1 {K1} gather {gather}: fused {timed(fused):.1f} us | two launches{' + gather' if gather else ''} {timed(two):.1f} us", flush=True)
