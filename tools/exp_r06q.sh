#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06q; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do for b in 128 64 256; do
  export DEVO_TRANSFORM_BLOCK=$b
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_$b" -o k -- python "$R/bench.py" --no-cpu-baseline --no-reference-api --no-full-iteration --no-f16 --steps 90 --warmup 9 > "$O/b_$b.json" 2> "$O/b_$b.err"
  echo "== block $b: $(python -c "import json;j=json.loads(open('$O/b_$b.json').read().strip().splitlines()[-1]);print(j['value'], j['ms_per_step'])") $(python "$R/tools/rocprof_summary.py" "$O/t_$b" 2>&1 | grep -E "k_transform" | awk '{print $3}')"
  rm -rf "$O/t_$b"
done; done
