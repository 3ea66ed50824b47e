#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel in libdevo_hip.so (north_star: "LDS / wavefront occupancy for fastba"):
compiler figures from `hipcc -Rpass-analysis=kernel-resource-usage` (VGPR, AGPR, SGPR, scratch, static LDS, waves/SIMD),
optionally joined with the LDS bytes per workgroup and the grid that a rocprofv3 kernel trace recorded
(dynamic LDS is only known at launch).   python tools/kernel_resources.py [trace_dir] [--json out.json]
--json: the LAUNCHED kernels of the trace as {base name: {vgpr, sgpr, scratch, lds_bytes (at launch), waves_per_simd (registers and LDS),
grid_wgs, wg_threads}} — bench.py's fastba report reads profiles/ba_kernel_resources.json.
(rocprofv3's own VGPR_Count column is not used: it does not match the kernel descriptors on gfx950.)"""
import csv, glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from devo_amd import build as B

json_out = None
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    json_out = sys.argv[i + 1]
    del sys.argv[i:i + 2]
records = {}
launch = {}
if len(sys.argv) > 1:
    for p in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            n = r["Kernel_Name"]
            if "devo::" in n and n not in launch:
                wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
                launch[re.sub(r"\(.*", "", n).replace("void ", "")] = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg, wg, int(r["LDS_Block_Size"]))
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>7} {'LDS static':>10} {'waves/SIMD':>10} | {'grid WGs':>8} {'WG':>5} {'LDS/WG at launch':>16} {'WGs/CU by LDS':>13}  kernel")
for src in B.SOURCES:
    cmd = [B._hipcc()] + B.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(B.CSRC, src), "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
            continue
        if cur is None:
            continue
        for key, tag in (("VGPRs:", "v"), ("AGPRs:", "a"), ("TotalSGPRs:", "s"), ("ScratchSize [bytes/lane]:", "sc"), ("Occupancy [waves/SIMD]:", "occ"), ("LDS Size [bytes/block]:", "lds")):
            if key in line and (key != "VGPRs:" or "AGPRs" not in line):
                cur[tag] = int(re.search(re.escape(key) + r"\s*(\d+)", line).group(1))
        if "lds" in cur:
            short = re.sub(r"\(.*", "", cur["name"]).replace("void ", "")
            g = launch.get(short)
            extra = f"{g[0]:8d} {g[1]:5d} {g[2]:16d} {min(32, 163840 // g[2]) if g[2] else 32:13d}" if g else f"{'-':>8} {'-':>5} {'-':>16} {'-':>13}"
            print(f"{cur.get('v', 0):5d} {cur.get('a', 0):5d} {cur.get('s', 0):5d} {cur.get('sc', 0):7d} {cur['lds']:10d} {cur.get('occ', 0):10d} | {extra}  {short[:80]}")
            if g:
                base = short.replace("devo::", "").split("<")[0]
                waves_wg = max(1, g[1] // 64)
                by_lds = (163840 // g[2]) * waves_wg // 4 if g[2] else 8           # resident waves per SIMD the launch's LDS allows
                records.setdefault(base, {"instance": short, "vgpr": cur.get("v", 0) + cur.get("a", 0), "sgpr": cur.get("s", 0), "scratch": cur.get("sc", 0),
                                          "lds_bytes": g[2], "waves_per_simd": max(1, min(cur.get("occ", 8), by_lds if by_lds > 0 else 1)), "grid_wgs": g[0], "wg_threads": g[1]})
            cur = None

if json_out:
    import json
    with open(json_out, "w") as f:
        json.dump(records, f, indent=1, sort_keys=True)
    print("wrote", json_out, len(records), "kernels")
