#!/usr/bin/env python3
"""Why does the fp32 lookup take 122 us inside the step and 111 us behind a launch of itself?  Graphs of different compositions, the lookup's
duration in each read from rocprofv3's kernel trace (run under rocprofv3 --kernel-trace; prints markers through the number of launches)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs
from devo_amd import synth
from devo_amd.backends import cuda_ba, cuda_corr

dev = torch.device("cuda", 0)
cfg = synth.workload("cfg2")
d, _ = build_inputs(cfg, 1234, dev, torch.float32, "blk8")
n, M, R = cfg["n"], cfg["M"], cfg["R"]
E = d["ii"].numel(); Np = d["patches"].shape[1]
ws = cuda_ba.workspace(E, Np, n - 1, dev)
Dm = 2 * R + 1
corr_out = torch.empty(1, E, Dm * Dm * 18, dtype=torch.float32, device=dev)
big = torch.empty(64 << 20, dtype=torch.float32, device=dev)          # 256 MB scratch for the "evict" variant
cuda_ba.prepare(d["kk"], Np, n - 1, ws)

def parts(which):
    if "restore" in which: torch.mul(d["state0"], 1.0, out=d["state"])
    coords, order = cuda_ba.transform(d["poses"], d["patches"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp", plan_for=(n, cfg["H"], R, cfg["W"], 0))
    order = cuda_corr.plan_finish(order, d["jj"], n, cfg["H"], R, width=cfg["W"], l1=0)
    cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], R, (1, 4), out=corr_out, order=order)
    if "ba" in which:
        cuda_ba.forward_delta(d["poses"], d["patches"], d["intr"], coords, d["delta"], d["weight"], d["lmbda"], d["ii"], d["jj"], d["kk"], 1, n, 2, ws)
    if "evict" in which: big[: (32 << 20)].add_(1.0)               # touch 128 MB (read + write) between lookups
    if "small" in which: big[: (1 << 18)].add_(1.0)                # a 1 MB elementwise kernel instead

names = ["restore+ba", "restore", "restore+evict", "restore+small"]
graphs = {}
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for nm in names:
        parts(nm); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(18): parts(nm)
        graphs[nm] = g
torch.cuda.current_stream().wait_stream(side)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for nm in names:
    g = graphs[nm]
    g.replay(); torch.cuda.synchronize()
    ev0.record()
    for _ in range(10): g.replay()
    ev1.record(); torch.cuda.synchronize()
    print(f"{nm:16s} {ev0.elapsed_time(ev1) / 180 * 1e3:8.1f} us per step", flush=True)
    # marker: a distinct number of tiny launches so that the trace can be split per variant
    torch.cuda.synchronize()
