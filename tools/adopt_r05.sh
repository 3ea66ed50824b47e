#!/bin/bash
# copy the artefacts of tools/collect_r05.sh (gpurun_out/r05) to profiles/r05_* and fold the counters into profiles/pmc_traffic.json
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r05; P=profiles
python tools/update_pmc_traffic.py $S r05 > /dev/null
cp $S/bench.json $P/r05_bench.json; cp $S/cfg2_kernel_trace.txt $P/r05_kernel_trace_bench_cfg2.txt; cp $S/stress_kernel_trace.txt $P/r05_kernel_trace_bench_stress.txt
for n in cfg2_f32 cfg2_f16 stress_f32; do cp $S/${n}_pmc_corr_fwd.txt $P/r05_${n}_pmc_corr_fwd.txt; done
cp $S/kernel_resources.txt $P/r05_kernel_resources.txt; cp $S/ba_kernel_resources.json $P/ba_kernel_resources.json; cp $S/stress_bench.json $P/r05_stress_bench.json
cp $S/reference_api_native.json $P/r05_reference_api_native.json; cp $S/reference_api_ctypes.json $P/r05_reference_api_ctypes.json
cp $S/group_form.txt $P/r05_group_form_final.txt; cp $S/mlp2.txt $P/r05_mlp2_final.txt; cp $S/update_op.txt $P/r05_update_op.txt; cp $S/corr_backward.txt $P/r05_corr_backward.txt
cp $S/ba_train_step.txt $P/r05_ba_train_step.txt; cp $S/train_mode.json $P/r05_train_mode.json; cp $S/train_categories.txt $P/r05_train_categories.txt
cp $S/cfg2_bench_under_rocprof.json $P/r05_cfg2_bench_under_rocprof.json; cp $S/rs_linear.txt $P/r05_rs_linear.txt; cp $S/update_f16_kernels.txt $P/r05_update_f16_kernels.txt
[ -f $S/rocprofv3_kernel_stats.csv ] && cp $S/rocprofv3_kernel_stats.csv $P/r05_rocprofv3_kernel_stats.csv
echo adopted
