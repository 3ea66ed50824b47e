#!/bin/bash
# Round 6, first GPU call: the fastba tests (old + round 6), the headline bench, a kernel trace of it.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_fastba.py tests/test_gpu_fastba_r06.py tests/test_abi.py -x -q -m gpu > "$O/pytest_fastba.txt" 2>&1
tail -5 "$O/pytest_fastba.txt"
timeout 600 python bench.py --no-cpu-baseline --no-reference-api > "$O/bench.json" 2> "$O/bench.err"
tail -c 1500 "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cfg2_trace" -o k -- python "$R/bench.py" --no-cpu-baseline --no-reference-api --no-full-iteration --steps 90 --warmup 9 > "$O/cfg2_bench_under_rocprof.json" 2> "$O/cfg2_trace.log"
python "$R/tools/rocprof_summary.py" "$O/cfg2_trace" > "$O/cfg2_kernel_trace.txt" 2>&1
rm -rf "$O/cfg2_trace"
head -24 "$O/cfg2_kernel_trace.txt"
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step")}, j["roofline"]["frac"], j["roofline"]["us_per_launch"], j["ba"]["gpu_ms"], j["ba"].get("gpu_ms_new_graph"), j["ba"].get("launches"), j.get("new_graph_every_step"), j.get("f16",{}).get("value"), j.get("update_op"))
print(j["ba"].get("kernels"))
PY
