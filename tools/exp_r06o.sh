#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06o; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for tag in old new old new; do
  if [ $tag = new ]; then export DEVO_LIB=$R/devo_amd/lib/libdevo_hip.so; else export DEVO_LIB=$R/devo_amd/lib/libdevo_$tag.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_$tag" -o k -- python "$R/tools/profile_ba.py" --reps 50 --workload stress > "$O/$tag.log" 2>&1
  echo "== $tag: $(grep 'BA ms' $O/$tag.log)"
  python "$R/tools/rocprof_summary.py" "$O/t_$tag" 2>&1 | grep -E "k_ba" | grep -v prepare | cut -c1-100
  rm -rf "$O/t_$tag"
done
