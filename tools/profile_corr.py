#!/usr/bin/env python3
"""Run only the altcorr lookup (both pyramid levels) a few times — the target of rocprofv3 --pmc passes."""
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
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--layout", default="blk8")
ap.add_argument("--no-plan", action="store_true")
ap.add_argument("--per-level", action="store_true", help="one launch per pyramid level instead of the fused two-level launch the bench uses")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = synth.workload(a.workload)
dt = torch.float32 if a.dtype == "f32" else torch.float16
d, _ = build_inputs(cfg, 1234, dev, dt, a.layout)
n, R = cfg["n"], cfg["R"]
E = d["ii"].numel()
Dm = 2 * R + 1
out = torch.empty(1, E, Dm * Dm * 18, dtype=dt, device=dev)
coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
order = None if a.no_plan else cuda_corr.plan(coords, d["jj"], n, cfg["H"], 1.0, R)
if a.no_plan:
    cuda_corr.PLAN_MIN_EDGES = 1 << 60
for _ in range(a.reps):
    if a.per_level:
        for lvl, (fm, s_) in enumerate(zip(d["pyramid"], (1.0, 4.0))):
            cuda_corr.forward_into(out, d["gmap"], fm, coords, d["kk"], d["jj"], R, Dm * Dm * 18, 2, lvl, order=order, coord_div=s_)
    else:
        cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], R, (1, 4), out=out, order=order)
torch.cuda.synchronize()
if order is not None:
    print("done", E, "heavy", int(order[E]), "dead", int(order[2 * E + 1]))
