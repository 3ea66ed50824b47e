#!/usr/bin/env python3
"""Time the event voxelisation + standardisation on the GPU: 1 M events into a 5 x 480 x 640 grid, and
std over a 15-frame sequence.  (The reference's CPU formulation, utils/event_utils.py:180-232 as restated in
oracle/events.py, takes 153 ms per 200 k events on the same host: measured once by the test infrastructure, not here.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devo_amd import events
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
N, H, W = 1_000_000, 480, 640
xs, ys = (torch.rand(N, generator=g) * W).floor(), (torch.rand(N, generator=g) * H).floor()
ts = torch.sort(torch.rand(N, generator=g, dtype=torch.float64) * 5e4).values
ps = torch.randint(0, 2, (N,), generator=g).to(torch.int8)
a = [t.to(dev) for t in (xs, ys, ts, ps)]
for _ in range(3): v = events.to_voxel_grid(*a, H, W, 5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): v = events.to_voxel_grid(*a, H, W, 5)
e1.record(); torch.cuda.synchronize()
print(f"to_voxel_grid  {N} events -> 5x{H}x{W}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
seq = v[None, None].repeat(1, 15, 1, 1, 1).contiguous()
for _ in range(3): events.std(seq)
torch.cuda.synchronize()
e0.record()
for _ in range(20): events.std(seq)
e1.record(); torch.cuda.synchronize()
print(f"std            [1,15,5,{H},{W}] ({seq.numel() * 4 / 1e6:.0f} MB): {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us (incl. the copy)")
