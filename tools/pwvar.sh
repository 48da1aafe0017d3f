export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# A/B of library builds / environment settings inside ONE GPU call: bench.py (100 steps) per variant
run() { env "$@" python bench.py --steps 100 --no-cpu-baseline --no-headline 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], {k:(v['ms_per_step'],v['frac']) for k,v in d['roofline']['kernels'].items() if k in ('pwconv','dwconv','pool','stem')})
except Exception as e: print('$*', 'failed', e)"; }
run FQ_X=0
for v in 4 6 12 16; do run FQ_DW_WG_PER_CU=$v; done
for v in 2 3 4; do run FQ_PWS_WG_PER_CU=$v; done
run FQ_X=0
