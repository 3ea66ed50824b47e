for d in 0 1 2 4 8 3 7 15 9; do echo "dbg=$d"; DEVO_LN_DBG=$d timeout 120 python tools/bench_linear_split.py 2>&1 | grep "^18000 x 384 x 384" | cut -c1-90; done
