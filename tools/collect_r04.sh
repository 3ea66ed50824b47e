#!/bin/bash
# Run on the GPU box (via gpurun): the round-4 artefacts — headline bench (json with the CPU baseline + kernel trace), FETCH / WRITE / TCC
# counters of the dense-product lookup kernel (cfg2 fp32 / fp16, stress), the stress bench, the reference-API probe, kernel resources,
# training mode.  Output: gpurun_out/<tag>/; tools/update_pmc_traffic.py folds the counters into profiles/pmc_traffic.json.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run_pmc() {     # name, extra profile_corr args
  local name=$1; shift
  local i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/pmc_${name}/p$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 "$@" > "$O/pmc_${name}_p$i.log" 2>&1
  done
  python "$R/tools/rocprof_summary.py" "$O/pmc_${name}" corr_fwd 2>&1 | sed "s#$O/##" > "$O/${name}_pmc_corr_fwd.txt"
}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cfg2_trace" -o k -- python "$R/bench.py" --no-cpu-baseline --no-reference-api --steps 100 --warmup 10 > "$O/cfg2_bench_under_rocprof.json" 2> "$O/cfg2_trace.log"
python "$R/tools/rocprof_summary.py" "$O/cfg2_trace" > "$O/cfg2_kernel_trace.txt" 2>&1
cp "$O"/cfg2_trace/*/*kernel_stats.csv "$O/rocprofv3_kernel_stats.csv" 2>/dev/null
run_pmc cfg2_f32
run_pmc cfg2_f16 --dtype f16
run_pmc stress_f32 --workload stress
python "$R/tools/kernel_resources.py" > "$O/kernel_resources.txt" 2>&1
timeout 900 python "$R/bench.py" --workload stress --steps 50 --warmup 5 --no-cpu-baseline > "$O/stress_bench.json" 2> "$O/stress_bench.err"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stress_trace" -o k -- python "$R/bench.py" --workload stress --steps 30 --warmup 3 --no-f16 --no-cpu-baseline > "$O/stress_bench_under_rocprof.json" 2> "$O/stress_trace.log"
python "$R/tools/rocprof_summary.py" "$O/stress_trace" > "$O/stress_kernel_trace.txt" 2>&1
DEVO_CORR_MM=0 timeout 400 python "$R/bench.py" --no-cpu-baseline --no-reference-api > "$O/mfma4x4_bench.json" 2> "$O/mfma4x4_bench.err"
timeout 600 python "$R/bench.py" --api reference > "$O/reference_api.json" 2> "$O/reference_api.err"
timeout 300 python "$R/tools/bench_corr_backward.py" 2>&1 | grep "per backward" > "$O/corr_backward.txt"
DEVO_BWD_NCHW=1 timeout 300 python "$R/tools/bench_corr_backward.py" 2>&1 | grep "per backward" >> "$O/corr_backward.txt"
timeout 300 python "$R/tools/bench_ba_train.py" > "$O/ba_train_step.txt" 2>&1
timeout 900 python "$R/bench.py" --mode train --steps 3 --warmup 1 > "$O/train_mode.json" 2> "$O/train_mode.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train_trace" -o k -- python "$R/bench.py" --mode train --steps 2 --warmup 1 > /dev/null 2> "$O/train_trace.log"
python "$R/tools/trace_categories.py" "$O/train_trace" > "$O/train_categories.txt" 2>&1
python "$R/tools/rocprof_summary.py" "$O/train_trace" 2>&1 | head -60 > "$O/train_kernel_trace.txt"
rm -rf "$O/train_trace" "$O/stress_trace"
timeout 900 python "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
tail -1 "$O/bench.json"
