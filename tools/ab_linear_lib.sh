#!/bin/bash
# Run on the GPU box: tools/bench_linear_split.py on several builds of the library (tools/build_variant.sh):  bash tools/ab_linear_lib.sh hip late ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for tag in "$@"; do echo "== $tag"; DEVO_LIB=$R/devo_amd/lib/libdevo_$tag.so timeout 200 python $R/tools/bench_linear_split.py 2>&1 | grep "^18000\|^4224" | cut -c1-110; done
