#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06u; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t" -o k -- python "$R/tools/profile_steady_host.py" > "$O/log.txt" 2>&1
python "$R/tools/rocprof_summary.py" "$O/t" 2>&1 | grep -E "calls|k_hash|k_neighbors|k_excl|k_kk|k_flag|k_rank|k_scatter|k_sort|fillBuffer|k_ba_prepare|softagg|k_rs_|corr_fwd|corr_order|corr_bin|elementwise|copyBuffer|gather" | cut -c1-125
rm -rf "$O/t"
