#!/bin/bash
# Run on the GPU box (via gpurun): the training step after the last changes of round 2 (split-K weight gradients, cached group maps,
# MIOpen DB) — json, per-iteration timing, kernel trace by category; MIOpen first-call cost with and without the shipped DB.
set -u
TAG=${1:-r02z}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$R/bench.py" --mode train --steps 3 --warmup 1 > "$O/train_mode.json" 2> "$O/train_mode.err"
timeout 300 python "$R/tools/bench_training_step.py" > "$O/training_step.txt" 2>&1
timeout 300 python "$R/tools/ubench_dw_gemm.py" 2>&1 | grep -v amdgpu > "$O/dw_gemm.txt"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train_trace" -o k -- python "$R/bench.py" --mode train --steps 2 --warmup 1 --train-iters 6 > /dev/null 2> "$O/train_trace.log"
python "$R/tools/rocprof_summary.py" "$O/train_trace" 2>&1 | head -40 > "$O/train_kernel_trace.txt"
python "$R/tools/miopen_first_call.py" 2>&1 | tail -1 | sed 's/^/shipped DB:  /' > "$O/miopen_first_call.txt"
MIOPEN_USER_DB_PATH=/tmp/mi_empty_cfg MIOPEN_CUSTOM_CACHE_DIR=/tmp/mi_empty_cache python "$R/tools/miopen_first_call.py" 2>&1 | tail -1 | sed 's/^/empty DB:    /' >> "$O/miopen_first_call.txt"
ls "$O"
