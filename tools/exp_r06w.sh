#!/bin/bash
# Round 6: SoftAgg grid sized from the group estimate — Update operator tests, timing, per-kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06w; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && timeout 900 python -m pytest tests/test_gpu_update.py tests/test_gpu_steady_state.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp
timeout 300 python $R/tools/bench_update.py > $O/update_op.txt 2>&1; grep "update op" $O/update_op.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/upd_trace -o k -- python $R/tools/bench_update.py --dtype f16 --only hip --reps 20 > /dev/null 2> $O/upd_trace.log
python $R/tools/rocprof_summary.py $O/upd_trace 2>&1 | head -14 | tee $O/update_f16_kernels.txt
rm -rf $O/upd_trace
