#!/usr/bin/env python3
"""csrc/linear.hip against the library's fp32 GEMM at the Update operator's training shapes.  python tools/bench_linear_split.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import devo_amd._lib as _L
if os.environ.get("DEVO_LIB"): _L.LIB_PATH = os.path.abspath(os.environ["DEVO_LIB"])      # a variant build (tools/build_variant.sh)
from devo_amd import update as U
dev = torch.device("cuda", 0)
def timed(fn, reps=50):
    """GPU time per call: the calls replayed from a HIP graph (no host time between launches)"""
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
if "--one" in sys.argv:                  # a few eager launches: the target of rocprofv3 --pmc passes (tools/pmc_linear.sh)
    x = torch.randn(18000, 384, device=dev); w = torch.randn(384, 384, device=dev) / 384 ** 0.5; b = torch.randn(384, device=dev)
    for _ in range(5): U._linear_split(x, w, b)
    torch.cuda.synchronize(); sys.exit(0)
for rows, n_out, k_in in [(18000, 384, 384), (18000, 768, 384), (18000, 384, 768), (96 * 22 * 2, 384, 384)]:
    x = torch.randn(rows, k_in, device=dev); w = torch.randn(n_out, k_in, device=dev) / k_in ** 0.5; b = torch.randn(n_out, device=dev)
    t_lib = timed(lambda: torch.nn.functional.linear(x, w, b))
    t_own = timed(lambda: U._linear_split(x, w, b))
    def resplit():
        U._wsplit_cache.clear(); U._split_weight(w, False)
    t_split = timed(resplit)
    U._split_weight(w, False)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    e_lib = (torch.nn.functional.linear(x, w, b).double() - ref).abs().max().item(); e_own = (U._linear_split(x, w, b).double() - ref).abs().max().item()
    fl = 2.0 * rows * n_out * k_in
    print(f"{rows} x {n_out} x {k_in}: library {t_lib:.1f} us ({fl / t_lib * 1e-6:.0f} TFLOP/s), split {t_own:.1f} us ({fl / t_own * 1e-6:.0f} TFLOP/s), "
          f"weight split {t_split:.1f} us; max |err| vs float64: library {e_lib:.2e}, split {e_own:.2e}")
print("fp16 storage (k_linear_f16) against the library:")
for rows, n_out, k_in in [(21600, 384, 384), (21600, 768, 384), (21600, 384, 882)]:
    x = torch.randn(rows, k_in, device=dev).half(); w = (torch.randn(n_out, k_in, device=dev) / k_in ** 0.5).half(); b = torch.randn(n_out, device=dev).half()
    t_lib = timed(lambda: torch.nn.functional.linear(x, w, b))
    t_own = timed(lambda: U._linear_f16(x, w, b))
    fl = 2.0 * rows * n_out * k_in
    print(f"{rows} x {n_out} x {k_in}: library {t_lib:.1f} us ({fl / t_lib * 1e-6:.0f} TFLOP/s), own {t_own:.1f} us ({fl / t_own * 1e-6:.0f} TFLOP/s)")
