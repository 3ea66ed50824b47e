#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace of the bench + separate PMC passes for the lookup kernel.
# Writes raw output under gpurun_out/<tag>/ and text summaries next to it; copy the summaries into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o k -- python "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/trace.log"
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $C --output-format csv -d "$O/pmc$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 > "$O/pmc$i.log" 2>&1
done
python "$R/tools/rocprof_summary.py" "$O/trace" > "$O/kernel_trace_summary.txt" 2>&1
python "$R/tools/rocprof_summary.py" "$O" corr_fwd 2>&1 | grep -v "kernel_trace.csv" > "$O/pmc_corr_fwd_summary.txt"
python "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
tail -1 "$O/bench.json"
