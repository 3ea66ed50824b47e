#!/usr/bin/env python3
"""Summarise rocprofv3 output (CSV kernel trace / counter collection) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict


def kernel_stats(path):
    agg = defaultdict(lambda: [0, 0.0])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name")
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            agg[name][0] += 1
            agg[name][1] += dur
            rows.append((int(r["Start_Timestamp"]), name, dur))
    # the lookup kernel by context: launches that follow a launch of the SAME kernel (bench.py's back-to-back graph: what roofline.us_per_launch
    # times) against launches inside a step (behind the reprojection / the BA kernels)
    rows.sort()
    ctx = defaultdict(lambda: [0, 0.0, 0, 0.0])
    for (_, prev, _), (_, name, dur) in zip(rows, rows[1:]):
        if "corr_fwd" in name:
            c = ctx[name]
            if prev == name:
                c[0] += 1; c[1] += dur
            else:
                c[2] += 1; c[3] += dur
    for name, c in ctx.items():
        if c[0] and c[2]:
            print(f"# {name[:70]}: {c[0]} launches behind a launch of the same kernel avg {c[1] / c[0]:.2f} us | {c[2]} launches inside steps avg {c[3] / c[2]:.2f} us")
    tot = sum(v[1] for v in agg.values()) or 1.0
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
    for name, (n, t) in rows[:25]:
        print(f"{n:7d} {t:12.1f} {t / n:10.2f} {100 * t / tot:6.2f}  {name[:110]}")


def kernel_resources(path, only="devo::"):
    """Launch geometry and occupancy limits of every kernel (north_star: LDS / wavefront occupancy for fastba):
    waves per SIMD allowed by VGPRs (512-entry file, granule 8) and workgroups per CU allowed by LDS (160 KiB)."""
    seen = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or ""
            if only not in name or name in seen:
                continue
            seen[name] = r
    print(f"{'grid(WGs)':>10} {'WG':>5} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'LDS B':>7} {'scratch':>7} {'waves/SIMD(vgpr)':>17} {'WGs/CU(lds)':>12}  kernel")
    for name, r in seen.items():
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
        v, a, lds = int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"])
        alloc = (v + a + 7) // 8 * 8
        wps = min(8, 512 // max(alloc, 8))
        wgs = 163840 // lds if lds else 32
        print(f"{grid:10d} {wg:5d} {v:5d} {a:5d} {int(r['SGPR_Count']):5d} {lds:7d} {int(r['Scratch_Size']):7d} {wps:17d} {min(wgs, 32):12d}  {name[:90]}")


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
        if match == "--resources":
            kernel_resources(p)
            continue
        kernel_stats(p)
    if match == "--resources":
        sys.exit(0)
    for p in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        print("==", p)
        counter_stats(p, match)
