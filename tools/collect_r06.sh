#!/bin/bash
# Run on the GPU box (via gpurun): the round-6 artefacts — headline bench under rocprofv3 (kernel trace + stats), FETCH / WRITE / TCC / TA counters of
# the lookup kernel (cfg2 fp32 / fp16, stress) in separate --pmc passes, kernel resources, the BA alone under the kernel trace, the stress bench, the
# reference-API probe (with the steady-state legs) through both bindings, the Update operator, training mode, the final bench line.
# Output: gpurun_out/<tag>/; tools/update_pmc_traffic.py folds the counters into profiles/pmc_traffic.json.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run_pmc() {     # name, extra profile_corr args
  local name=$1; shift
  local i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_TA_BUSY_sum SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/pmc_${name}/p$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 "$@" > "$O/pmc_${name}_p$i.log" 2>&1
  done
  python "$R/tools/rocprof_summary.py" "$O/pmc_${name}" corr_fwd 2>&1 | sed "s#$O/##" > "$O/${name}_pmc_corr_fwd.txt"
  rm -rf "$O/pmc_${name}"
}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cfg2_trace" -o k -- python "$R/bench.py" --no-cpu-baseline --no-reference-api --steps 90 --warmup 9 > "$O/cfg2_bench_under_rocprof.json" 2> "$O/cfg2_trace.log"
python "$R/tools/rocprof_summary.py" "$O/cfg2_trace" > "$O/cfg2_kernel_trace.txt" 2>&1
find "$O/cfg2_trace" -name "*kernel_stats.csv" -exec cp {} "$O/rocprofv3_kernel_stats.csv" \; 2>/dev/null
python "$R/tools/kernel_resources.py" "$O/cfg2_trace" --json "$O/ba_kernel_resources.json" > "$O/kernel_resources.txt" 2>&1
rm -rf "$O/cfg2_trace"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ba_trace" -o k -- python "$R/tools/profile_ba.py" --reps 300 > "$O/ba_alone.log" 2>&1
python "$R/tools/rocprof_summary.py" "$O/ba_trace" 2>&1 | head -12 > "$O/ba_kernel_trace.txt"; grep "BA ms" "$O/ba_alone.log" >> "$O/ba_kernel_trace.txt"
rm -rf "$O/ba_trace"
run_pmc cfg2_f32
run_pmc cfg2_f16 --dtype f16
run_pmc stress_f32 --workload stress
timeout 900 python "$R/bench.py" --workload stress --steps 54 --warmup 5 --no-cpu-baseline > "$O/stress_bench.json" 2> "$O/stress_bench.err"
timeout 600 python "$R/bench.py" --api reference > "$O/reference_api_native.json" 2> "$O/reference_api.err"
DEVO_BINDING=ctypes timeout 600 python "$R/bench.py" --api reference > "$O/reference_api_ctypes.json" 2>> "$O/reference_api.err"
timeout 300 python "$R/tools/bench_update.py" > "$O/update_op.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/upd_trace" -o k -- python "$R/tools/bench_update.py" --dtype f16 --only hip --reps 20 > /dev/null 2> "$O/upd_trace.log"
python "$R/tools/rocprof_summary.py" "$O/upd_trace" 2>&1 | head -14 > "$O/update_f16_kernels.txt"
rm -rf "$O/upd_trace"
timeout 300 python "$R/tools/bench_ba_train.py" > "$O/ba_train_step.txt" 2>&1
timeout 900 python "$R/bench.py" --mode train --steps 3 --warmup 1 > "$O/train_mode.json" 2> "$O/train_mode.err"
timeout 600 python "$R/tools/profile_train_sections.py" 2>&1 | grep -v "amdgpu\|Warning\|warn" | tail -14 > "$O/train_sections.txt"
timeout 900 python "$R/bench.py" --with-stress > "$O/bench.json" 2> "$O/bench.err"
tail -1 "$O/bench.json" | cut -c1-400
