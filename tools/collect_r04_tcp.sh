#!/bin/bash
# Run on the GPU box (via gpurun): round-4 evidence on what caps the lookup's intake per CU —
# (1) tools/ubench/l2_fill (fill rate by access shape, L2-resident / L1-resident), (2) TCP / TA / TCC counters of the shipped
# per-edge lookup kernel for cfg2 fp32 and fp16 (separate --pmc passes, nothing but the counters in each run).
set -u
TAG=${1:-r04tcp}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 "$R/tools/ubench/l2_fill" > "$O/l2_fill.txt" 2>&1
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TA|TD|TCC)_[A-Z0-9_]+(_sum)?\b" | sort -u | tr '\n' ' ' > "$O/counter_names.txt"
pass() {   # tag, dtype args..., then counters
  local tag=$1 dt=$2; shift 2
  local i=0
  for C in "$@"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/pmc_${tag}/p$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 $dt > "$O/pmc_${tag}_p$i.log" 2>&1
  done
  python "$R/tools/rocprof_summary.py" "$O/pmc_${tag}" corr_fwd 2>&1 | sed "s#$O/##" > "$O/${tag}_pmc_corr_fwd.txt"
}
CS=("TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" "TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE")
pass f32 "" "${CS[@]}"
pass f16 "--dtype f16" "${CS[@]}"
for f in "$O"/pmc_*_p*.log; do if grep -q -i "error\|invalid\|not found" "$f"; then echo "== $f"; grep -i -m3 "error\|invalid\|not found" "$f"; fi; done > "$O/pmc_errors.txt"
cat "$O/l2_fill.txt"
cat "$O/f32_pmc_corr_fwd.txt"
