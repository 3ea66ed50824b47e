#!/usr/bin/env python3
"""cuda_ba.prepare (unique patches, edges grouped by patch) alone: the cfg2 full graph (kk ascending) and DEVO's steady-state sliding-window graph
(45 312 edges in devo.py's order, patch slots of a 2048-frame buffer)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from devo_amd import synth
from devo_amd.backends import cuda_ba
dev = "cuda"
_g = torch.Generator().manual_seed(0)
_kk_cfg2 = synth.full_graph(15, 96)[2]
_kk_win30 = synth.sliding_window_graph(30, 96)[2]
cases = {"cfg2 graph, edges shuffled (E = 21 600)": (_kk_cfg2[torch.randperm(len(_kk_cfg2), generator=_g)], 15 * 96, 14),
         f"sliding window after 30 keyframes (E = {len(_kk_win30)}, devo.py's order), 2048-frame buffers": (_kk_win30, 2048 * 96, 10),
         "sliding window after 12 keyframes (devo.py's order)": (synth.sliding_window_graph(12, 96)[2], 2048 * 96, 10),
         "cfg2 full graph (E = 21 600, ascending kk)": (synth.full_graph(15, 96)[2], 15 * 96, 14),
         "sliding window after 40 keyframes (E = 45 312, devo.py's order), 48-frame buffers": (synth.sliding_window_graph(40, 96)[2], 48 * 96, 10),
         "the same, 2048-frame buffers (196 608 patch slots)": (synth.sliding_window_graph(40, 96)[2], 2048 * 96, 10)}
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, (kk, Np, N) in cases.items():
    kk = kk.to(dev)
    ws = cuda_ba.workspace(kk.numel(), Np, N, dev)
    for _ in range(3): cuda_ba.prepare(kk, Np, N, ws)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(50): cuda_ba.prepare(kk, Np, N, ws)
    ev1.record(); torch.cuda.synchronize()
    n, kx, seg, perm = cuda_ba.prepared_tables(ws, kk.numel(), Np, N)
    ref = torch.unique(kk)
    assert n == ref.numel() and torch.equal(kx.long(), ref)
    print(f"{name}: {ev0.elapsed_time(ev1) / 50 * 1e3:.1f} us per prepare")
    import ctypes
    from devo_amd import _lib as L
    if hasattr(L.lib(), "devo_debug_prep_trace"):
        buf = (ctypes.c_ulonglong * 16)()
        L.lib().devo_debug_prep_trace(buf)
        st = [buf[i] for i in range(9)]
        names = ["kk loaded + ascending test", "range + flags", "ids ranked (scan)", "segments counted", "segment starts (scan)", "scattered", "starts published", "segments sorted"]
        print("   phases (us): " + ", ".join(f"{n} {(st[i + 1] - st[i]) / 100.0:.1f}" for i, n in enumerate(names) if st[i + 1] > st[i]))
