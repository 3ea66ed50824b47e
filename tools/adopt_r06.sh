#!/bin/bash
# Copy the artefacts of tools/collect_r06.sh (gpurun_out/<tag>/) into profiles/ under their round-6 names and fold the counters into pmc_traffic.json.
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/${1:-r06}
cp $S/bench.json profiles/r06_bench.json
cp $S/cfg2_bench_under_rocprof.json profiles/r06_cfg2_bench_under_rocprof.json
cp $S/cfg2_kernel_trace.txt profiles/r06_kernel_trace_bench_cfg2.txt
cp $S/rocprofv3_kernel_stats.csv profiles/r06_rocprofv3_kernel_stats.csv
cp $S/kernel_resources.txt profiles/r06_kernel_resources.txt
cp $S/ba_kernel_resources.json profiles/ba_kernel_resources.json
cp $S/ba_kernel_trace.txt profiles/r06_ba_kernel_trace.txt
for n in cfg2_f32 cfg2_f16 stress_f32; do cp $S/${n}_pmc_corr_fwd.txt profiles/r06_${n}_pmc_corr_fwd.txt; done
cp $S/stress_bench.json profiles/r06_stress_bench.json
cp $S/reference_api_native.json profiles/r06_reference_api_native.json
cp $S/reference_api_ctypes.json profiles/r06_reference_api_ctypes.json
cp $S/update_op.txt profiles/r06_update_op.txt
cp $S/update_f16_kernels.txt profiles/r06_update_f16_kernels.txt
cp $S/ba_train_step.txt profiles/r06_ba_train_step.txt
cp $S/train_mode.json profiles/r06_train_mode.json
cp $S/train_sections.txt profiles/r06_train_sections.txt
python tools/update_pmc_traffic.py $S r06
