export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
for g in 0 1 0 1; do python bench.py --steps 200 --no-cpu-baseline --no-headline --graph $g 2>&1 | tail -1 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print('graph $g', d['value'], d['ms_per_step'])
except Exception as e: print('failed', t[-600:])"; done
