#!/bin/bash
# Round 6, Update operator: does a second co-resident workgroup per CU pay for streaming the weights twice?  Single 384 x 384 layer of the
# row-resident kernel at 96 / 64 / 48 / 32 rows per workgroup (1 / 1 / 2 / 2 workgroups per CU), 21 600 rows.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
for mt in 6 4 3 2; do
  echo "== DEVO_RS_MT=$mt" >> $O/exp_r06v.txt
  DEVO_RS_MT=$mt timeout 120 python $R/tools/bench_rs.py 2>&1 | grep "^rows" >> $O/exp_r06v.txt
  DEVO_RS_MT=$mt DEVO_RS_TRACE=1 timeout 120 python $R/tools/bench_rs.py 2>&1 | grep "rs trace" | head -3 >> $O/exp_r06v.txt
done
cat $O/exp_r06v.txt
