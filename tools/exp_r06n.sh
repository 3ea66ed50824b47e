#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06n; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
python "$R/tools/exp_instep.py" 2>&1 | grep -v amdgpu
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$O/t" -o k -- python "$R/tools/exp_instep.py" > "$O/log.txt" 2>&1
python - <<PY
import csv, glob, collections
rows=[]
for fn in glob.glob("$O/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# lookup durations grouped by the name of the kernel that ran right before the lookup's predecessor chain: classify by what follows the lookup
out=collections.defaultdict(list)
for i,(s,e,nm) in enumerate(rows):
    if "corr_fwd_mm" in nm and i+1 < len(rows):
        nxt = rows[i+1][2]
        key = "ba" if "k_ba_acc" in nxt else ("evict/small/restore:" + nxt[:60])
        out[key].append((e-s)/1000.0)
for k,v in out.items():
    v=sorted(v); print(f"{len(v):5d} lookups followed by {k:70s} median {v[len(v)//2]:7.2f} us  mean {sum(v)/len(v):7.2f}")
PY
rm -rf "$O/t"
