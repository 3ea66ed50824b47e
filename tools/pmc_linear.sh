#!/bin/bash
# Run on the GPU box: SQ / TA counters of k_linear_split at 18000 x 384 x 384 (tools/bench_linear_split.py --one).  Output: gpurun_out/<tag>/
set -u
TAG=${1:-pmc_linear}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "TA_TA_BUSY_sum TA_BUSY_avr SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/p$i" -o p -- python "$R/tools/bench_linear_split.py" --one > "$O/p$i.log" 2>&1
done
python "$R/tools/rocprof_summary.py" "$O" k_linear_split 2>&1 | sed "s#$O/##" > "$O/k_linear_split_pmc.txt"
