#!/usr/bin/env python3
"""Fold the counter summaries of tools/collect_r04.sh (gpurun_out/<tag>/<name>_pmc_corr_fwd.txt) into profiles/pmc_traffic.json:
    python tools/update_pmc_traffic.py gpurun_out/r04 r04"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
recs = json.load(open(path))
for name, key in (("cfg2_f32", "cfg2/f32/blk8/fused/mm"), ("cfg2_f16", "cfg2/f16/blk8/fused/mm"), ("stress_f32", "stress/f32/blk8/fused/mm")):
    f = os.path.join(src, f"{name}_pmc_corr_fwd.txt")
    if not os.path.exists(f):
        continue
    vals, kernel = {}, None
    for line in open(f):
        m = re.match(r"\s+(\w+)\s+dispatches\s+\d+\s+avg/dispatch\s+([0-9.]+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
        elif "corr_fwd" in line:
            kernel = re.search(r"(corr_fwd_\w+_kernel)", line).group(1)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        recs[key] = {"round": tag, "kernel": kernel, "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                     "TCC_MISS_sum": vals.get("TCC_MISS_sum"), "TCC_HIT_sum": vals.get("TCC_HIT_sum"), "TCC_REQ_sum": vals.get("TCC_REQ_sum"),
                     "TCP_TCC_READ_REQ_sum": vals.get("TCP_TCC_READ_REQ_sum")}
        print(key, recs[key])
json.dump(recs, open(path, "w"), indent=1)
