#!/usr/bin/env python3
"""Static instruction mix of a kernel from hipcc's assembly (hipcc ... -S --cuda-device-only -o x.s): per basic block and in total.
    python tools/isa_mix.py x.s <substring of the mangled kernel name> [--blocks]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith(("buffer_load", "global_load", "flat_load")): return "vmem_ld"
    if op.startswith(("buffer_", "global_", "flat_")): return "vmem_st"
    if op.startswith("ds_"): return "lds"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m and key in m.group(1):
            name = m.group(1)
            tot = collections.Counter()
            blocks = []
            cur, curc = "entry", collections.Counter()
            i += 1
            while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
                ln = lines[i].strip()
                i += 1
                if not ln or ln.startswith((";", "//")):
                    continue
                lm = re.match(r"^(\.LBB\w+):", ln)
                if lm:
                    blocks.append((cur, curc)); cur, curc = lm.group(1), collections.Counter()
                    continue
                if ln.startswith("."):
                    continue
                op = ln.split()[0]
                c = classify(op)
                tot[c] += 1; curc[c] += 1
                if c == "branch":
                    curc["->" + ln.split()[-1]] += 1
            blocks.append((cur, curc))
            print(name[:90])
            print("   total:", dict(sorted(tot.items())))
            if show_blocks:
                for b, c in blocks:
                    if sum(v for k, v in c.items() if not k.startswith("->")) >= 8:
                        print(f"   {b:12s}", dict(sorted(c.items())))
        i += 1


main()
