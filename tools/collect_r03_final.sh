#!/bin/bash
# Run on the GPU box (via gpurun): the last artefacts of round 3, after the backward and the BA changes that followed tools/collect_r03.sh —
# default bench line (with the CPU baseline), kernel trace of the bench, kernel resources, training mode, the backward per level.
set -u
TAG=${1:-r03g}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cfg2_trace" -o k -- python "$R/bench.py" --no-cpu-baseline --steps 100 --warmup 10 > "$O/cfg2_bench_under_rocprof.json" 2> "$O/cfg2_trace.log"
python "$R/tools/rocprof_summary.py" "$O/cfg2_trace" > "$O/cfg2_kernel_trace.txt" 2>&1
python "$R/tools/kernel_resources.py" > "$O/kernel_resources.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bwd_trace" -o k -- python "$R/tools/bench_corr_backward.py" > "$O/corr_backward.txt" 2>&1
python "$R/tools/rocprof_summary.py" "$O/bwd_trace" 2>&1 | head -8 > "$O/corr_backward_kernel_trace.txt"
timeout 300 python "$R/tools/bench_corr_backward.py" 1.0 2>&1 | grep "per backward" >> "$O/corr_backward.txt"
DEVO_CORR_BWD_ATOMIC=1 timeout 300 python "$R/tools/bench_corr_backward.py" 2>&1 | grep "per backward" >> "$O/corr_backward.txt"
timeout 900 python "$R/bench.py" --mode train --steps 3 --warmup 1 > "$O/train_mode.json" 2> "$O/train_mode.err"
timeout 900 python "$R/bench.py" --workload stress --steps 50 --warmup 5 --no-cpu-baseline > "$O/stress_bench.json" 2> "$O/stress_bench.err"
timeout 900 python "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
tail -1 "$O/bench.json"
