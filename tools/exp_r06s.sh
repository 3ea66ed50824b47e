#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06s; rm -rf "$O"; mkdir -p "$O"; cd "$R"

cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t" -o k -- python "$R/bench.py" --api reference > "$O/ref.json" 2> "$O/ref.err"
python "$R/tools/rocprof_summary.py" "$O/t" 2>&1 | head -60 | cut -c1-130
python - <<PY
import json
j=json.loads(open("$O/ref.json").read().strip().splitlines()[-1]); j=j.get("reference_api", j)
print({k:v.get("ms_per_iter", v.get("ms_per_frame")) for k,v in j.items() if isinstance(v,dict)})
PY
rm -rf "$O/t"
