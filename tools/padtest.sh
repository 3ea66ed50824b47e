R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for pad in 0 64 128 512 2048; do
  DEVO_PLANE_PAD=$pad timeout 300 python $R/bench.py --no-cpu-baseline --no-train-probe --no-reference-api --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('pad $pad', 'f32 it/s', d['value'], 'lookup us', d['roofline']['us_per_launch'], 'b2b', d['roofline']['us_per_launch_back_to_back'], '| f16 it/s', d['f16']['value'], 'lookup us', d['f16']['roofline']['us_per_launch'], 'b2b', d['f16']['roofline']['us_per_launch_back_to_back'])"
done
