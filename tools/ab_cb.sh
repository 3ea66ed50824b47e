for cb in 8 4 16; do for dt in f32 f16; do echo "cb=$cb $dt"; DEVO_BENCH_CB=$cb timeout 300 python bench.py --no-cpu-baseline --no-reference-api --no-f16 --dtype $dt --steps 100 --warmup 10 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(' ', j['value'], 'it/s  lookup', j['roofline']['us_per_launch'], 'us')
    elif 'rror' in l: print(l.strip()[:200])
"; done; done
