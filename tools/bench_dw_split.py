#!/usr/bin/env python3
"""csrc/linear_dw.hip against the library's products for dW / db at the Update operator's training shapes.  python tools/bench_dw_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import update as U
dev = torch.device("cuda", 0)
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rows, no, ni in [(18000, 384, 384), (18000, 768, 384), (18000, 384, 768)]:
    dy = torch.randn(rows, no, device=dev); x = torch.randn(rows, ni, device=dev)
    S = 16
    def lib():
        gw = torch.bmm(dy.reshape(S, rows // S, -1).transpose(1, 2), x.reshape(S, rows // S, -1)).sum(0)
        return gw, dy.sum(0)
    t_lib = timed(lib)
    t_direct = timed(lambda: dy.t() @ x)
    t_own = timed(lambda: U._dw_split(dy, x, True))
    ref = dy.double().t() @ x.double()
    e_lib = (lib()[0].double() - ref).abs().max().item(); e_own = (U._dw_split(dy, x, True)[0].double() - ref).abs().max().item()
    print(f"{rows} rows, {no} x {ni}: library (16 row chunks + sum + bias sum) {t_lib:.1f} us, library direct {t_direct:.1f} us, split {t_own:.1f} us "
          f"(DEVO_DW_SPLITS={os.environ.get('DEVO_DW_SPLITS', 'auto')}); max |err| vs float64: library {e_lib:.2e}, split {e_own:.2e}")
