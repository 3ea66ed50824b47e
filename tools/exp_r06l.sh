#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_update.py tests/test_gpu_devo_iteration.py -x -q -m gpu 2>&1 | tail -3
mkdir -p gpurun_out/r06; timeout 300 python tools/bench_update.py > gpurun_out/r06/update_op.txt 2>&1; tail -12 gpurun_out/r06/update_op.txt
