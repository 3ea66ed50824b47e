#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06p; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for wl in cfg2 stress; do
for tag in old hip tw kp old hip tw kp; do
  export DEVO_LIB=$R/devo_amd/lib/libdevo_$tag.so
  reps=300; [ $wl = stress ] && reps=50
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_$tag" -o k -- python "$R/tools/profile_ba.py" --reps $reps --workload $wl > "$O/$tag.log" 2>&1
  echo "== $wl $tag: $(grep 'BA ms' $O/$tag.log) | $(python "$R/tools/rocprof_summary.py" "$O/t_$tag" 2>&1 | grep -E "k_ba_(acc|solve|reduce)" | awk '{printf "%s %s | ", $3, substr($6,1,28)}')"
  rm -rf "$O/t_$tag"
done; done
