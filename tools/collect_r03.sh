#!/bin/bash
# Run on the GPU box (via gpurun): the round-3 artefacts — headline bench (json + kernel trace), FETCH / WRITE / TCC counters of the
# lookup (cfg2 fp32 / fp16, stress) for the default per-edge kernel and for the opt-in region-shared kernel, stress benches,
# the BA solver's phase stamps and the micro-benchmarks its design rests on, kernel resources.
# Output: gpurun_out/<tag>/; copy the summaries into profiles/.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run_bench() {   # name, extra bench args
  local name=$1; shift
  timeout 400 python "$R/bench.py" --no-cpu-baseline "$@" > "$O/${name}_bench.json" 2> "$O/${name}_bench.err"
}
run_trace() {   # name, extra bench args
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${name}_trace" -o k -- python "$R/bench.py" --no-cpu-baseline --steps 100 --warmup 10 "$@" > "$O/${name}_bench_under_rocprof.json" 2> "$O/${name}_trace.log"
  python "$R/tools/rocprof_summary.py" "$O/${name}_trace" > "$O/${name}_kernel_trace.txt" 2>&1
}
run_pmc() {     # name, extra profile_corr args
  local name=$1; shift
  local i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/pmc_${name}/p$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 "$@" > "$O/pmc_${name}_p$i.log" 2>&1
  done
  python "$R/tools/rocprof_summary.py" "$O/pmc_${name}" corr_fwd 2>&1 | sed "s#$O/##" > "$O/${name}_pmc_corr_fwd.txt"
}
# headline
run_trace cfg2
run_pmc cfg2_f32
run_pmc cfg2_f16 --dtype f16
run_bench stress --workload stress --steps 50 --warmup 5
run_trace stress --workload stress --steps 30 --warmup 3 --no-f16
run_pmc stress_f32 --workload stress
# the opt-in region-shared kernel
DEVO_CORR_REGION=1 run_bench region_cfg2
DEVO_CORR_REGION=1 run_bench region_stress --workload stress --steps 30 --warmup 3
DEVO_CORR_REGION=1 run_pmc region_cfg2_f32
DEVO_CORR_REGION=1 run_pmc region_cfg2_f16 --dtype f16
# more SQ counters of both kernels (fp16, where they differ most)
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/sq_peredge/p$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 --dtype f16 > "$O/sq_peredge_p$i.log" 2>&1
  DEVO_CORR_REGION=1 timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/sq_region/p$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 --dtype f16 > "$O/sq_region_p$i.log" 2>&1
done
python "$R/tools/rocprof_summary.py" "$O/sq_peredge" corr_fwd 2>&1 | sed "s#$O/##" > "$O/sq_peredge_f16_pmc.txt"
python "$R/tools/rocprof_summary.py" "$O/sq_region" corr_fwd 2>&1 | sed "s#$O/##" > "$O/sq_region_f16_pmc.txt"
# BA solver
for m in 1 5; do DEVO_BA_TRACE=$m timeout 300 python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-f16 --no-train-probe 2>&1 | grep -m3 "ba trace" | sed "s/^/DEVO_BA_TRACE=$m  /"; done > "$O/ba_solver_stamps.txt"
for m in 1; do DEVO_BA_SOLVE_V1=1 DEVO_BA_TRACE=$m timeout 300 python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-f16 --no-train-probe 2>&1 | grep -m3 "ba trace" | sed "s/^/DEVO_BA_SOLVE_V1=1 DEVO_BA_TRACE=$m  /"; done >> "$O/ba_solver_stamps.txt"
DEVO_BA_SOLVE_V1=1 run_bench solver_v1 --no-f16
for u in wave_sync chain_step tile_step; do echo "== tools/ubench/$u"; timeout 120 "$R/tools/ubench/$u" 2>&1 | grep -v amdgpu; done > "$O/ba_solver_ubench.txt"
python "$R/tools/kernel_resources.py" > "$O/kernel_resources.txt" 2>&1
# training mode, default bench with the CPU baseline
timeout 900 python "$R/bench.py" --mode train --steps 3 --warmup 1 > "$O/train_mode.json" 2> "$O/train_mode.err"
timeout 900 python "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
tail -1 "$O/bench.json"
ls "$O"
