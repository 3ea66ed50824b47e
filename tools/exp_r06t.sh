#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06t; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
python "$R/bench.py" --api reference 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); j=j.get('reference_api', j)
print(j['f16_steady_state_frame_with_update_operator'])"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t" -o k -- python "$R/bench.py" --api reference > "$O/ref.json" 2> "$O/ref.err"
python "$R/tools/rocprof_summary.py" "$O/t" 2>&1 | grep -E "calls|prepare|k_kk|k_flag|k_excl|k_rank|k_scatter|k_sort|hash|neighbors|fillBuffer|copyBuffer|FillFunctor|k_rs_|softagg" | cut -c1-120
rm -rf "$O/t"
