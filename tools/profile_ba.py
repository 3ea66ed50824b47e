#!/usr/bin/env python3
"""Run only the bundle adjustment (2 GN iterations) a few times — target of rocprofv3 passes."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs
from devo_amd import synth
from devo_amd.backends import cuda_ba
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = synth.workload(a.workload)
cfg2 = dict(cfg); cfg2["C"] = 16
d, _ = build_inputs(cfg2, 1234, dev, torch.float32, "cl")
n = cfg["n"]
E = d["ii"].numel()
ws = cuda_ba.workspace(E, d["patches"].shape[1], n - 1, dev)
coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
tgt = coords[:, :, :, 1, 1] + d["delta"]
def run():
    d["state"].copy_(d["state0"])
    cuda_ba.forward(d["poses"], d["patches"], d["intr"], tgt, d["weight"], d["lmbda"], d["ii"], d["jj"], d["kk"], 1, n, 2, ws=ws)
run(); torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(a.reps): run()
ev1.record(); torch.cuda.synchronize()
print("BA ms", ev0.elapsed_time(ev1) / a.reps)
