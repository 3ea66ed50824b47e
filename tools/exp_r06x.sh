#!/bin/bash
# Round 6: the next lookup's locality plan on the BA's solver launch — tests, the bench line (plan_in_line beside it), kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06x; rm -rf $O; mkdir -p $O
cd $R && timeout 900 python -m pytest tests/test_gpu_fastba_r06.py tests/test_gpu_fastba.py tests/test_abi.py tests/test_gpu_binding_legs.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 900 python bench.py --no-cpu-baseline --no-reference-api > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "plan_in_line", "new_graph_every_step")})
print("f16", {k: d["f16"].get(k) for k in ("value", "ms_per_step")})
print(d["roofline"]["us_per_launch"], d["roofline"]["us_per_launch_back_to_back"], d["ba"]["gpu_ms"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python $R/bench.py --no-cpu-baseline --no-reference-api --no-f16 --no-full-iteration --steps 90 --warmup 9 > $O/bench_under_rocprof.json 2> $O/trace.log
python $R/tools/rocprof_summary.py $O/trace 2>&1 | head -16 | tee $O/kernel_trace.txt
rm -rf $O/trace
