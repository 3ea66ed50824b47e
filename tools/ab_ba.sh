#!/bin/bash
# A/B of libdevo_<tag>.so builds on the BA alone (rocprofv3 kernel trace of tools/profile_ba.py), alternating, twice: tools/ab_ba.sh "cfg2 stress" old hip ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab_ba; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
WLS=$1; shift
for wl in $WLS; do for rep in 1 2; do for tag in "$@"; do
  export DEVO_LIB=$R/devo_amd/lib/libdevo_$tag.so
  reps=300; [ $wl = stress ] && reps=50
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_$tag" -o k -- python "$R/tools/profile_ba.py" --reps $reps --workload $wl > "$O/$tag.log" 2>&1
  echo "== $wl $tag: $(grep 'BA ms' $O/$tag.log | cut -c1-14) | $(python "$R/tools/rocprof_summary.py" "$O/t_$tag" 2>&1 | grep -E "k_ba_(acc|solve|reduce)" | awk '{printf "%s %s | ", $3, substr($6,1,24)}')"
  rm -rf "$O/t_$tag"
done; done; done
