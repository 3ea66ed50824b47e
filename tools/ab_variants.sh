#!/bin/bash
# Run on the GPU box: bench.py on several builds of the library (tools/build_variant.sh) — in-step lookup time, fp32 and fp16.
#   bash tools/ab_variants.sh hip e2 e4 ...      (hip = the default build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  DEVO_LIB=$R/devo_amd/lib/libdevo_$tag.so timeout 300 python $R/tools/bench_with_lib.py --no-train-probe --no-reference-api --no-full-iteration --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag', 'f32 it/s', d['value'], 'lookup us', d['roofline']['us_per_launch'], 'b2b', d['roofline']['us_per_launch_back_to_back'], '| f16 it/s', d['f16']['value'], 'lookup us', d['f16']['roofline']['us_per_launch'], 'b2b', d['f16']['roofline']['us_per_launch_back_to_back'])"
done
