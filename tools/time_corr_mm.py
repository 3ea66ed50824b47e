#!/usr/bin/env python3
"""A/B of the fused two-level lookup: dense-product kernel (corr_mm.h) against the 4x4 matrix-core kernel (corr_mfma.h) on the
bench's inputs — max relative difference between the two, against the fp64 oracle on a sample of edges, and HIP-event times."""
import argparse
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs                                  # noqa: E402
from devo_amd import synth                                      # noqa: E402
from devo_amd.backends import cuda_ba, cuda_corr                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--layouts", default="blk8")
ap.add_argument("--oracle-edges", type=int, default=300)
ap.add_argument("--C", type=int, default=0, help="override the number of feature channels (experiment: fp16 at C = 256 has the load / product count of an fp32 pyramid stored as fp16 hi + lo planes)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = synth.workload(a.workload)
if a.C:
    cfg = dict(cfg, C=a.C)
for dt in (torch.float16, torch.float32):
    for layout in a.layouts.split(","):
        d, cpu = build_inputs(cfg, 1234, dev, dt, layout)
        n, R = cfg["n"], cfg["R"]
        E = d["ii"].numel()
        Dm = 2 * R + 1
        coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
        order = cuda_corr.plan(coords, d["jj"], n, cfg["H"], 1.0, R, width=cfg["W"], l1=0)
        res = {}
        for name, mm in (("mfma4x4", False), ("dense", True)):
            cuda_corr.MM_KERNEL = mm
            out = torch.zeros(1, E, Dm * Dm * 18, dtype=dt, device=dev)
            for _ in range(3):
                cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], R, (1, 4), out=out, order=order)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
            ev[0].record()
            for i in range(a.reps):
                cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], R, (1, 4), out=out, order=order)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(a.reps))
            res[name] = (out.float().clone(), ts[len(ts) // 2], ts[0])
        ref, got = res["mfma4x4"][0], res["dense"][0]
        scale = ref.abs().max().item()
        print(f"{a.workload} {str(dt).split('.')[-1]} {layout}: mfma4x4 median {res['mfma4x4'][1]:.1f} us (min {res['mfma4x4'][2]:.1f}) | dense median "
              f"{res['dense'][1]:.1f} us (min {res['dense'][2]:.1f}) | max |dense - mfma4x4| / scale = {(got - ref).abs().max().item() / scale:.3e}", flush=True)
        if a.oracle_edges:
            from oracle import altcorr as A
            k = min(a.oracle_edges, E)
            sel = torch.randperm(E, generator=torch.Generator().manual_seed(1))[:k]
            q = (lambda t: t.to(dt).float())                          # the oracle sees the stored (rounded) values
            f0 = q(cpu["fmap"]); f1 = q(synth.pyramid_l1(cpu["fmap"])); g = q(cpu["gmap"])
            cs = coords[:, sel.to(dev)].cpu()
            kk_, jj_ = d["kk"][sel.to(dev)].cpu(), d["jj"][sel.to(dev)].cpu()
            r0 = A.corr_forward(g, f0, cs, kk_, jj_, R)
            r1 = A.corr_forward(g, f1, cs / 4, kk_, jj_, R)
            rr = torch.stack([r0, r1], -1).view(1, k, -1)
            for name in res:
                err = (res[name][0][:, sel.to(dev)].cpu() - rr).abs().max().item() / rr.abs().max().item()
                print(f"    {name}: max error against the oracle on {k} edges / scale = {err:.3e}")
