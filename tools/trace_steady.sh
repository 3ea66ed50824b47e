#!/bin/bash
# kernel trace of the steady-state frame loop (tools/profile_steady_cprofile.py: 40 frames) -> per-frame GPU time by kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tr; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o st -- python $R/tools/profile_steady_cprofile.py > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
python - "$f" > "$O/steady_kernel_stats.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
out = []
for r in rows:
    n, t = int(r["Calls"]), float(r["TotalDurationNs"])
    out.append((t / 40 / 1e3, n / 40, r["Name"][:110]))
    tot += t
print(f"GPU time per frame {tot / 40 / 1e3:.1f} us (40 frames, warm-up included)")
for t, n, name in sorted(out, reverse=True)[:45]:
    print(f"{t:8.1f} us  {n:5.1f} calls  {name}")
PY
cat "$O/steady_kernel_stats.txt"
# the kernel sequence of ONE steady frame (the 35th), with start offsets: where the GPU idles
python - "$(find /tmp/tr -name '*kernel_trace.csv' | head -1)" > "$O/steady_frame_sequence.txt" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# frames are delimited by k_transform<true> launches
idx = [i for i, r in enumerate(rows) if "k_transform" in r["Kernel_Name"]]
a, b = idx[-5], idx[-4]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
print(f"frame of {b - a} kernels, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us between transforms")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:5.1f}  {r['Kernel_Name'][:100]}")
    prev_end = e
PY
