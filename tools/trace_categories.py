#!/usr/bin/env python3
"""Kernel time of a rocprofv3 --kernel-trace --stats run by category (hipBLASLt GEMMs, MIOpen, this repo's kernels, ATen families).
python tools/trace_categories.py <trace dir>"""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = [(int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, r["Name"]) for r in csv.DictReader(open(f))]
tot = sum(r[1] for r in rows)
def cat(n):
    if n.startswith("Cijk"): return "hipBLASLt GEMMs"
    if "devo::" in n or n.startswith("k_ba_edge") : return "this repo's HIP kernels"
    if n.startswith("igemm") or n.startswith("miopen") or "Sp3AsmConv" in n or "gridwise" in n.lower() or "naive_conv" in n or "SubTensor" in n \
            or "batched_transpose" in n or "Im2d2Col" in n or "Col2Im" in n or "transpose_" in n or "xdlops" in n: return "MIOpen convolutions"
    if "layer_norm" in n or "GammaBeta" in n: return "ATen layer norm"
    if "instance_norm" in n or "batch_norm" in n or "BatchNorm" in n: return "ATen instance norm"
    if "scatter" in n or "gather" in n or "index" in n.lower(): return "ATen gather / scatter / index"
    if "reduce_kernel" in n: return "ATen reductions"
    if "copyBuffer" in n or "fillBuffer" in n or "FillFunctor" in n or "direct_copy" in n: return "copies / fills"
    if "elementwise" in n: return "ATen elementwise"
    if "sort" in n.lower() or "rocprim" in n or "unique" in n.lower(): return "sort / scan"
    return "other"
cats = {}
for c, t, n in rows:
    k = cat(n); cats[k] = (cats.get(k, (0, 0))[0] + c, cats.get(k, (0, 0))[1] + t)
print(f"kernel time by category ({os.path.basename(os.path.normpath(d))}; total {tot / 1e3:.1f} ms, {sum(r[0] for r in rows)} launches)")
for k, (c, v) in sorted(cats.items(), key=lambda x: -x[1][1]):
    print(f"  {k:32s} {v / 1e3:8.1f} ms {100 * v / tot:5.1f} %  {c:6d} launches")
for c, t, n in sorted([r for r in rows if cat(r[2]) == "this repo's HIP kernels"], key=lambda x: -x[1])[:10]:
    print(f"    {c:5d} calls {t / 1e3:7.2f} ms  {n[:100]}")
