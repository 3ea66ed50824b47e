#!/bin/bash
# h layer of the SoftAgg over the patches at the steady-state graph's size (2 112 - 4 000 groups): 32-row or 96-row workgroups?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; rm -f $O/exp_r06y.txt
for rows in 1440 2112 3072 4096 6144 8192 12288; do
  for sm in 2048 16384; do
    echo "rows $rows DEVO_RS_SMALL_M=$sm: $(DEVO_RS_SMALL_M=$sm timeout 120 python $R/tools/bench_rs.py $rows 2>&1 | grep '^rows' | head -1 | cut -c1-90)" >> $O/exp_r06y.txt
  done
done
cat $O/exp_r06y.txt
