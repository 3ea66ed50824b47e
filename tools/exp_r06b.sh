#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06b; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gpu_fastba_r06.py -x -q -m gpu 2>&1 | tail -3
DEVO_LIB=devo_amd/lib/libdevo_acctrace.so timeout 120 python tools/acc_trace.py 2>&1 | grep -v amdgpu.ids | tee "$O/acc_trace.txt"
DEVO_BA_TRACE=1 timeout 120 python tools/profile_ba.py --reps 3 2>&1 | grep "ba trace" | tail -4 | tee "$O/solve_trace.txt"
DEVO_BA_TRACE=7 timeout 120 python tools/profile_ba.py --reps 1 2>&1 | grep -A16 "arrival" | tail -17 | tee -a "$O/solve_trace.txt"
for i in 1 2 3; do timeout 120 python tools/profile_ba.py --reps 200 2>&1 | grep "BA ms"; done
