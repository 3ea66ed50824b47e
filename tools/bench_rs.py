#!/usr/bin/env python3
"""csrc/gemm_rs.hip (rows in LDS once, weights from the L2 into registers) against csrc/linear.hip's k_linear_f16: identical results,
HIP-graph replays, us per layer at the update operator's shapes.  DEVO_RS_MT = 4 / 6 / 8 row tiles per workgroup."""
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


torch.manual_seed(0)
for N, relu_from, with_res in ((384, None, False), (384, 0, True), (768, 384, False)):
    lin = torch.nn.Linear(384, N).to(dev).half()
    x = (torch.randn(rows, 384, device=dev) * 0.5).half()
    res = torch.randn(rows, N, device=dev).half() if with_res else None
    out = {}
    for rs in (True, False):
        UA.RS_GEMM = rs
        r = res.clone() if with_res else None
        y = UA._linear_f16(x, lin.weight, lin.bias, relu_from=relu_from, residual=r, out=r)
        out[rs] = y.clone()
        t = timed(lambda: UA._linear_f16(x, lin.weight, lin.bias, relu_from=relu_from, residual=r, out=r))
        out[("t", rs)] = t
    ref = torch.nn.functional.linear(x.float(), lin.weight.float(), lin.bias.float())
    if relu_from is not None:
        ref[:, relu_from:].relu_()
    if with_res:
        ref += res.float()
    err = (out[True].float() - ref).abs().max().item()
    same = torch.equal(out[True], out[False])
    print(f"rows {rows} K 384 N {N} relu_from {relu_from} residual {with_res}: rs {out[('t', True)]:.1f} us | linear.hip {out[('t', False)]:.1f} us | "
          f"identical {same} | max err vs fp32 {err:.2e}", flush=True)
