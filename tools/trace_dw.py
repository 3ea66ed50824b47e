#!/usr/bin/env python3
"""Cycle stamps of workgroup (0, 0), wave 0 of k_dw_split (a -DDW_TRACE build: tools/build_variant.sh dwtrace linear_dw -DDW_TRACE):
per step [requests issued | 64 values per operand read (+ maxima, column sums) | scales checked | scaled + split | products issued | tiles landed + barrier]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import devo_amd._lib as _L
_L.LIB_PATH = os.path.abspath(os.environ.get("DEVO_LIB", "devo_amd/lib/libdevo_dwtrace.so"))
from devo_amd import update as U
dev = torch.device("cuda", 0)
for rows in (640, 18000):
    dy = torch.randn(rows, 384, device=dev); x = torch.randn(rows, 384, device=dev)
    for _ in range(3): U._dw_split(dy, x, True)
    torch.cuda.synchronize()
    t = next(iter(U._dw_ws.values()))[:40].cpu().long().tolist()          # (one workspace per (device, stream): this script uses one)
    print(f"rows {rows}: prologue {t[1]}; steps:", " ".join("[" + " ".join(str(t[2 + 6 * i + j] - t[1 + 6 * i + j]) for j in range(6)) + "]" for i in range(5)))
