#!/bin/bash
# Run on the GPU box: kernel trace of the training step (bench.py --mode train), by category and by kernel.  Output: gpurun_out/<tag>/
set -u
TAG=${1:-train}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train_trace" -o k -- python "$R/bench.py" --mode train --steps 2 --warmup 1 > /dev/null 2> "$O/train_trace.log"
python "$R/tools/trace_categories.py" "$O/train_trace" > "$O/train_categories.txt" 2>&1
python "$R/tools/rocprof_summary.py" "$O/train_trace" 2>&1 | head -90 > "$O/train_kernel_trace.txt"
rm -rf "$O/train_trace"
timeout 600 python "$R/bench.py" --mode train --steps 3 --warmup 2 > "$O/train_mode.json" 2> "$O/train_mode.err"
