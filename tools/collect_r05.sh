#!/bin/bash
# Run on the GPU box (via gpurun): the round-5 artefacts — headline bench under rocprofv3 (kernel trace + stats), FETCH / WRITE / TCC / TA
# counters of the lookup kernel (cfg2 fp32 / fp16, stress) in separate --pmc passes, kernel resources (table + the JSON bench.py's fastba
# report reads), the stress bench, the reference-API probe through both bindings, the group form against the per-edge form, the fused MLP
# micro-benchmark, training mode, the final bench line.  Output: gpurun_out/<tag>/; tools/update_pmc_traffic.py folds the counters into
# profiles/pmc_traffic.json.
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run_pmc() {     # name, extra profile_corr args
  local name=$1; shift
  local i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_TA_BUSY_sum SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/pmc_${name}/p$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 "$@" > "$O/pmc_${name}_p$i.log" 2>&1
  done
  python "$R/tools/rocprof_summary.py" "$O/pmc_${name}" corr_fwd 2>&1 | sed "s#$O/##" > "$O/${name}_pmc_corr_fwd.txt"
  rm -rf "$O/pmc_${name}"
}
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cfg2_trace" -o k -- python "$R/bench.py" --no-cpu-baseline --no-reference-api --steps 100 --warmup 10 > "$O/cfg2_bench_under_rocprof.json" 2> "$O/cfg2_trace.log"
python "$R/tools/rocprof_summary.py" "$O/cfg2_trace" > "$O/cfg2_kernel_trace.txt" 2>&1
find "$O/cfg2_trace" -name "*kernel_stats.csv" -exec cp {} "$O/rocprofv3_kernel_stats.csv" \; 2>/dev/null
python "$R/tools/kernel_resources.py" "$O/cfg2_trace" --json "$O/ba_kernel_resources.json" > "$O/kernel_resources.txt" 2>&1
rm -rf "$O/cfg2_trace"
run_pmc cfg2_f32
run_pmc cfg2_f16 --dtype f16
run_pmc stress_f32 --workload stress
timeout 900 python "$R/bench.py" --workload stress --steps 50 --warmup 5 --no-cpu-baseline > "$O/stress_bench.json" 2> "$O/stress_bench.err"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stress_trace" -o k -- python "$R/bench.py" --workload stress --steps 30 --warmup 3 --no-f16 --no-cpu-baseline > "$O/stress_bench_under_rocprof.json" 2> "$O/stress_trace.log"
python "$R/tools/rocprof_summary.py" "$O/stress_trace" > "$O/stress_kernel_trace.txt" 2>&1
rm -rf "$O/stress_trace"
timeout 600 python "$R/bench.py" --api reference > "$O/reference_api_native.json" 2> "$O/reference_api.err"
DEVO_BINDING=ctypes timeout 600 python "$R/bench.py" --api reference > "$O/reference_api_ctypes.json" 2>> "$O/reference_api.err"
timeout 300 python "$R/tools/time_group.py" > "$O/group_form.txt" 2>&1
timeout 300 python "$R/tools/bench_mlp2.py" > "$O/mlp2.txt" 2>&1
timeout 300 python "$R/tools/bench_update.py" > "$O/update_op.txt" 2>&1
DEVO_UPD_RS_CHAINS=0 timeout 300 python "$R/tools/bench_update.py" --dtype f16 --only hip 2>&1 | grep "update op" | sed "s/$/   (DEVO_UPD_RS_CHAINS=0)/" >> "$O/update_op.txt"
DEVO_UPD_RS_CHAINS=0 DEVO_UPD_RS=0 timeout 300 python "$R/tools/bench_update.py" --dtype f16 --only hip 2>&1 | grep "update op" | sed "s/$/   (DEVO_UPD_RS_CHAINS=0 DEVO_UPD_RS=0: round 5 before the row-resident kernels)/" >> "$O/update_op.txt"
DEVO_UPD_RS_SPLIT=0 timeout 300 python "$R/tools/bench_update.py" --dtype f32 --only hip 2>&1 | grep "update op" | sed "s/$/   (DEVO_UPD_RS_SPLIT=0)/" >> "$O/update_op.txt"
timeout 300 python "$R/tools/bench_rs.py" > "$O/rs_linear.txt" 2>&1
DEVO_RS_TRACE=1 timeout 200 python "$R/tools/bench_update.py" --dtype f16 --only hip --reps 2 2>&1 | grep "gru trace" | tail -1 >> "$O/rs_linear.txt"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/upd_trace" -o k -- python "$R/tools/bench_update.py" --dtype f16 --only hip --reps 20 > /dev/null 2> "$O/upd_trace.log"
python "$R/tools/rocprof_summary.py" "$O/upd_trace" 2>&1 | head -14 > "$O/update_f16_kernels.txt"
rm -rf "$O/upd_trace"
timeout 300 python "$R/tools/bench_corr_backward.py" 2>&1 | grep "per backward" > "$O/corr_backward.txt"
timeout 300 python "$R/tools/bench_ba_train.py" > "$O/ba_train_step.txt" 2>&1
timeout 900 python "$R/bench.py" --mode train --steps 3 --warmup 1 > "$O/train_mode.json" 2> "$O/train_mode.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train_trace" -o k -- python "$R/bench.py" --mode train --steps 2 --warmup 1 > /dev/null 2> "$O/train_trace.log"
python "$R/tools/trace_categories.py" "$O/train_trace" > "$O/train_categories.txt" 2>&1
rm -rf "$O/train_trace"
timeout 900 python "$R/bench.py" --with-stress > "$O/bench.json" 2> "$O/bench.err"
tail -1 "$O/bench.json" | cut -c1-600
