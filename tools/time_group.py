#!/usr/bin/env python3
"""Fused two-level lookup on the bench's inputs: group form (level 1 from LDS regions, group plan) against the per-edge form (edge plan) —
HIP-event medians, fp32 and fp16.  DEVO_LIB=<path> times another build of the library (tools/build_variant.sh)."""
import argparse
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import devo_amd._lib as L                                        # noqa: E402
if os.environ.get("DEVO_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["DEVO_LIB"])
from bench import build_inputs                                  # noqa: E402
from devo_amd import synth                                      # noqa: E402
from devo_amd.backends import cuda_ba, cuda_corr                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--kinds", default="edges,groups")
ap.add_argument("--dtypes", default="f32,f16")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = synth.workload(a.workload)
for dn in a.dtypes.split(","):
    dt = torch.float32 if dn == "f32" else torch.float16
    d, cpu = build_inputs(cfg, 1234, dev, dt, "blk8")
    n, R = cfg["n"], cfg["R"]
    E = d["ii"].numel()
    Dm = 2 * R + 1
    coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
    line = []
    ref = None
    for kind in a.kinds.split(","):
        order = cuda_corr.plan(coords, d["jj"], n, cfg["H"], 1.0, R, width=cfg["W"] if kind == "groups" else 0, l1=4 if kind == "groups" else 0)
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
        same = "" if ref is None else (" identical" if torch.equal(out, ref) else " DIFFERENT")
        ref = out if ref is None else ref
        o = order.cpu()
        line.append(f"{kind} {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f}){same} [heavy {int(o[E])} dead {int(o[2 * E + 1])}]")
    print(f"{os.path.basename(L.LIB_PATH)} {a.workload} {dn}: " + " | ".join(line), flush=True)
