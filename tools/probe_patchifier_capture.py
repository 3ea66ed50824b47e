#!/usr/bin/env python3
"""Which pieces of the Patchifier's inference call can be captured into a HIP graph (and what a replay of the encoders costs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import patchifier as PF
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pf = PF.Patchifier().to(dev).eval()
images = torch.randn(1, 1, 5, 480, 640, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    pf(images, 96, scorer_eval_mode="topk")
lp = pf._lowp_modules(torch.float16)
x16 = images.half()
def attempt(name, fn, reps=30):
    try:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / reps * 1e3
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
            with torch.cuda.graph(g, stream=s):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): g.replay()
        torch.cuda.synchronize()
        print(f"{name:34s} eager {te:7.3f} ms   graph replay {(time.perf_counter() - t0) / reps * 1e3:7.3f} ms")
    except Exception as ex:
        print(f"{name:34s} capture failed: {type(ex).__name__}: {str(ex)[:90]}")
        torch.cuda.synchronize()
with torch.no_grad():
    attempt("fnet (low-precision copy)", lambda: lp["fnet"](x16))
    attempt("inet", lambda: lp["inet"](x16))
    attempt("scorer", lambda: torch.sigmoid(lp["scorer"](x16).float()))
    attempt("fnet + inet + scorer", lambda: (lp["fnet"](x16), lp["inet"](x16), torch.sigmoid(lp["scorer"](x16).float())))
    smap = torch.sigmoid(lp["scorer"](x16).float())
    attempt("select topk", lambda: PF.select(smap, 96, "topk", True))
    attempt("select multi", lambda: PF.select(smap, 96, "multi", True))
    def whole():
        with torch.autocast("cuda", dtype=torch.float16):
            return pf(images, 96, scorer_eval_mode="topk")
    attempt("whole call (topk)", whole)
