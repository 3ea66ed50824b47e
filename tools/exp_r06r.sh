#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06r; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t" -o k -- python "$R/bench.py" --api reference > "$O/ref.json" 2> "$O/ref.err"
python "$R/tools/rocprof_summary.py" "$O/t" 2>&1 | head -40 | cut -c1-150
rm -rf "$O/t"
