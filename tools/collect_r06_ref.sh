#!/bin/bash
# the reference-API probes alone (both bindings) into gpurun_out/r06/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p "$O"; cd "$R"
timeout 600 python bench.py --api reference > "$O/reference_api_native.json" 2> "$O/reference_api.err"
DEVO_BINDING=ctypes timeout 600 python bench.py --api reference > "$O/reference_api_ctypes.json" 2>> "$O/reference_api.err"
timeout 300 python tools/bench_prepare.py 2>&1 | grep -v amdgpu > "$O/prepare.txt"
timeout 300 python tools/profile_steady_host.py 2>&1 | grep -v amdgpu > "$O/steady_host.txt"
cat "$O/prepare.txt" "$O/steady_host.txt"
python - <<PY
import json
for b in ("native","ctypes"):
    j=json.loads(open("$O/reference_api_%s.json" % b).read().strip().splitlines()[-1]); j=j.get("reference_api", j)
    print(b, {k:(v.get("it_per_s") or v.get("frames_per_s"), v.get("ms_per_iter") or v.get("ms_per_frame")) for k,v in j.items() if isinstance(v,dict)})
PY
