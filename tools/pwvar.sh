# A/B of library builds / environment settings inside ONE GPU call: bench.py (100 steps) per variant
run() { env "$@" python bench.py --steps 100 --no-cpu-baseline --no-headline 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], {k:(v['ms_per_step'],v['frac']) for k,v in d['roofline']['kernels'].items() if k in ('pwconv','dwconv','pool','stem')})
except Exception as e: print('$*', 'failed', e)"; }
L=quantization/mxnet_amd/csrc/build
run FQ_PW_FORM=0
run FQ_LIB_PATH=$L/lib_pwt2.so
run FQ_PW_FORM=0
run FQ_LIB_PATH=$L/lib_pwt2.so
