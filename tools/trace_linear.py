#!/usr/bin/env python3
"""Cycle stamps of workgroup 0 of k_linear_split (DEVO_LN_DBG=16): prologue, then per K step
[next step fetched, requests issued | next step checked | products + next split issued | loads landed | barrier passed]."""
import os, sys
os.environ["DEVO_LN_DBG"] = "16"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import update as U
dev = torch.device("cuda", 0)
for rows in (128, 4224, 18000):
    x = torch.randn(rows, 384, device=dev); w = torch.randn(384, 384, device=dev) / 384 ** 0.5; b = torch.randn(384, device=dev)
    for _ in range(3): y = U._linear_split(x, w, b)
    torch.cuda.synchronize()
    t = y[0, :32].cpu().long().tolist()
    print(f"rows {rows}: prologue {t[1]}; steps (issue, split, products, wait, barrier):", " ".join("[" + " ".join(str(t[2 + 5 * i + j] - t[1 + 5 * i + j]) for j in range(5)) + "]" for i in range(6)))
