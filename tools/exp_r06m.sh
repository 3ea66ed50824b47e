#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06m; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for per in 1024 512 340 256; do
  export DEVO_ORDER_EDGES_PER_WG=$per
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_$per" -o k -- python "$R/bench.py" --no-cpu-baseline --no-reference-api --no-full-iteration --no-f16 --steps 90 --warmup 9 > "$O/b_$per.json" 2> "$O/b_$per.err"
  echo "== edges per order workgroup $per: $(python -c "import json;j=json.loads(open('$O/b_$per.json').read().strip().splitlines()[-1]);print(j['value'], j['ms_per_step'])")"
  python "$R/tools/rocprof_summary.py" "$O/t_$per" 2>&1 | grep -E "corr_order|k_prepare_and" | cut -c1-90
  rm -rf "$O/t_$per"
done
