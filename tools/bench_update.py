#!/usr/bin/env python3
"""Time the Update operator at the metric's configuration (cfg2: E = 21 600 edges, dim 384):
HIP inference path (library GEMMs + fused kernels of csrc/update.hip) vs the plain torch composition of the same module
on the same GPU, fp32 and fp16 storage.  Not part of the headline metric (SURVEY §8d excludes the Update MLP)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devo_amd import synth
from devo_amd.update import Update

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="", help="hip | torch")
ap.add_argument("--dtype", default="", help="f32 | f16")
a = ap.parse_args()
cfg = synth.workload(a.workload)
dev = torch.device("cuda", 0)
ii, jj, kk = (t.to(dev) for t in synth.full_graph(cfg["n"], cfg["M"]))
E = ii.numel()
torch.manual_seed(0)
for dtype in [d for d in (torch.float32, torch.float16) if not a.dtype or a.dtype == {torch.float32: 'f32', torch.float16: 'f16'}[d]]:
    m = Update(3).to(dev).to(dtype).eval()
    net, inp = torch.randn(1, E, 384, device=dev, dtype=dtype), torch.randn(1, E, 384, device=dev, dtype=dtype)
    corr = torch.randn(1, E, 882, device=dev, dtype=dtype)
    def hip():
        with torch.no_grad():
            return m(net, inp, corr, None, ii, jj, kk)
    def ref():
        with torch.no_grad():
            return m.forward_torch(net, inp, corr, ii, jj, kk)
    def hip_autocast():                                             # devo.py:311: the fp32 operator called under autocast
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return m(net, inp, corr, None, ii, jj, kk)
    for name, fn in [(n_, f_) for n_, f_ in (("hip", hip), ("torch", ref)) + ((("hip, autocast", hip_autocast),) if dtype == torch.float32 else ())
                     if not a.only or a.only == n_.split(",")[0]]:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"update op  E={E} dim=384 {str(dtype)[6:]:8s} {name:14s} {e0.elapsed_time(e1) / a.reps:8.3f} ms", flush=True)
        if name.startswith("hip"):
            # the same call replayed from a HIP graph (how a captured DEVO update step runs it): the eager figure above is bound by
            # the host's launch rate (~25 launches), this one by the device
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(3): graph.replay()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.reps): graph.replay()
            e1.record(); torch.cuda.synchronize()
            print(f"update op  E={E} dim=384 {str(dtype)[6:]:8s} {name + ', graph replay':18s} {e0.elapsed_time(e1) / a.reps:8.3f} ms", flush=True)
