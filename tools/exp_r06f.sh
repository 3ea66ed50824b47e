#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06f; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_steady_state.py tests/test_abi.py tests/test_reference_binding.py -x -q -m gpu 2>&1 | tail -25
timeout 600 python bench.py --api reference > "$O/reference_api.json" 2> "$O/reference_api.err"; tail -3 "$O/reference_api.err"
python - <<PY
import json
j=json.loads(open("$O/reference_api.json").read().strip().splitlines()[-1])
for k,v in j.items():
    if isinstance(v, dict): print(k, v)
    elif "error" in k: print(k, v)
PY
