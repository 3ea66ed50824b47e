#!/usr/bin/env python3
"""Kernel sequence of ONE differentiable Gauss-Newton step (forward + backward) from a rocprofv3 kernel trace of tools/bench_ba_train.py:
   rocprofv3 --kernel-trace --output-format csv -d DIR -o k -- python tools/bench_ba_train.py;  python tools/trace_ba_train_kernels.py DIR"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda x: x[0])
names = [r[2] for r in rows]
# the fused path's steps: find the last two occurrences of the edge-terms kernel (one per step) and print what lies between
idx = [i for i, n in enumerate(names) if "k_ba_edge_terms" in n and "bwd" not in n and "backward" not in n]
sel = None
for a, b in zip(idx[:-1], idx[1:]):
    seg = names[a:b]
    if any("solve_terms" in n or "k_bt_" in n or "vjp" in n for n in seg) and any("vjp" in n for n in seg):
        sel = (a, b)
if sel is None:
    sel = (idx[-2], idx[-1])
a, b = sel
a = max(0, a - 1)            # the transform kernel in front of the edge terms
print(f"{b - a} launches between two consecutive steps (forward + backward):")
t0 = rows[a][0]
for s, e, n in rows[a:b]:
    print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  {n[:110]}")
