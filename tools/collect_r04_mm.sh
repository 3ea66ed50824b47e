#!/bin/bash
# Run on the GPU box (via gpurun): the dense-product lookup kernel (corr_mm.h) against the 4x4 matrix-core kernel — A/B times and
# differences (tools/time_corr_mm.py), per-wave phase stamps (DEVO_CORR_TRACE=1), SQ / TA / TCP counters of the dense kernel.
set -u
TAG=${1:-r04mm}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$R/tools/time_corr_mm.py" 2>&1 | grep -v amdgpu.ids > "$O/ab.txt"
for dt in f16 f32; do
  DEVO_CORR_TRACE=1 timeout 300 python "$R/tools/profile_corr.py" --reps 2 --dtype $dt 2>&1 | grep "trace" | sed "s/^/$dt  /" >> "$O/trace.txt"
done
if [ "${2:-pmc}" = "pmc" ]; then
for dt in f16 f32; do
  i=0
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/pmc_$dt/p$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 --dtype $dt > "$O/pmc_${dt}_p$i.log" 2>&1
  done
  python "$R/tools/rocprof_summary.py" "$O/pmc_$dt" corr_fwd 2>&1 | sed "s#$O/##" | grep -v "^==" > "$O/${dt}_pmc_corr_fwd.txt"
done
fi
cat "$O/ab.txt" "$O/trace.txt"
