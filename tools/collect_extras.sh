#!/bin/bash
# Run on the GPU box (via gpurun): the measurements DESIGN.md quotes beyond the headline — stress configuration, the NCHW-fp16
# drop-in path, training step, Update operator, event voxelisation — each as a JSON / text artefact (+ rocprofv3 kernel traces
# and FETCH/WRITE counters for the lookup-bound ones).  Output: gpurun_out/<tag>/; copy the summaries into profiles/.
set -u
TAG=${1:-r02x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run_bench() {   # name, extra bench args
  local name=$1; shift
  timeout 400 python "$R/bench.py" --no-cpu-baseline --no-f16 "$@" > "$O/${name}_bench.json" 2> "$O/${name}_bench.err"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${name}_trace" -o k -- python "$R/bench.py" --no-cpu-baseline --no-f16 --steps 50 --warmup 5 "$@" > /dev/null 2> "$O/${name}_trace.log"
  python "$R/tools/rocprof_summary.py" "$O/${name}_trace" > "$O/${name}_kernel_trace.txt" 2>&1
}
run_pmc() {     # name, extra profile_corr args
  local name=$1; shift
  local i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/${name}_pmc$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 "$@" > "$O/${name}_pmc$i.log" 2>&1
  done
  python "$R/tools/rocprof_summary.py" "$O" corr_fwd 2>&1 | grep -v "kernel_trace.csv" > "$O/${name}_pmc_corr_fwd.txt"
}
run_bench stress --workload stress --steps 50 --warmup 5
run_bench stress_f16 --workload stress --dtype f16 --steps 50 --warmup 5
run_bench nchw_f16 --layout nchw --dtype f16 --steps 50 --warmup 5
run_bench nchw_f32 --layout nchw --steps 50 --warmup 5
run_bench cfg2_f16 --dtype f16
run_pmc stress --workload stress
run_pmc cfg2_f16 --dtype f16
run_pmc nchw_f16 --layout nchw --dtype f16
timeout 300 python "$R/tools/bench_training_step.py" > "$O/training_step.txt" 2>&1
timeout 300 python "$R/tools/bench_update.py" > "$O/update.txt" 2>&1
timeout 300 python "$R/tools/bench_events.py" > "$O/events.txt" 2>&1
timeout 600 python "$R/bench.py" --mode train --steps 3 --warmup 1 > "$O/train_mode.json" 2> "$O/train_mode.err"
ls "$O"
