#!/usr/bin/env python3
"""Summarise rocprofv3 output (CSV kernel trace / counter collection) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict


def kernel_stats(path):
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name")
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            agg[name][0] += 1
            agg[name][1] += dur
    tot = sum(v[1] for v in agg.values()) or 1.0
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
    for name, (n, t) in rows[:25]:
        print(f"{n:7d} {t:12.1f} {t / n:10.2f} {100 * t / tot:6.2f}  {name[:110]}")


def counter_stats(path, match):
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if match not in k:
                continue
            agg[k.split("(")[0][:60]][r["Counter_Name"]][0] += 1
            agg[k.split("(")[0][:60]][r["Counter_Name"]][1] += float(r["Counter_Value"])
    for k, cs in agg.items():
        print(k)
        for c, (n, v) in sorted(cs.items()):
            print(f"    {c:28s} dispatches {n:5d}  avg/dispatch {v / n:16.1f}")


if __name__ == "__main__":
    d = sys.argv[1]
    match = sys.argv[2] if len(sys.argv) > 2 else "corr_fwd"
    for p in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        print("==", p)
        kernel_stats(p)
    for p in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        print("==", p)
        counter_stats(p, match)
