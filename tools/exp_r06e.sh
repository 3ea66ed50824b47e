#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06e; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for g in 1 4 12 23 45 90 180; do
  DEVO_BA_RETRACT_WGS=$g timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_$g" -o k -- python "$R/tools/profile_ba.py" --reps 200 > "$O/$g.log" 2>&1
  echo "== G=$g: $(grep 'BA ms' $O/$g.log) $(python "$R/tools/rocprof_summary.py" "$O/t_$g" 2>&1 | grep -E "k_ba_solve" | cut -c1-60)"
  rm -rf "$O/t_$g"
done
DEVO_BA_FUSE_RETRACT=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_u" -o k -- python "$R/tools/profile_ba.py" --reps 200 > "$O/u.log" 2>&1
echo "== unfused: $(grep 'BA ms' $O/u.log)"; python "$R/tools/rocprof_summary.py" "$O/t_u" 2>&1 | grep -E "k_ba_" | cut -c1-80
