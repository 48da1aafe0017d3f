export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
for r in 1 2; do for m in 4096 3000 1500 700; do
  FQ_PWS_THIN_MIN_TILES=$m python3 bench.py --model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline --steps 200 --no-cpu-baseline --no-headline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']
print('FQ_PWS_THIN_MIN_TILES=$m:', l['value'], 'images/s; one batch at a time', l['single_stream']['value'], {n:round(v['ms_per_step'],3) for n,v in k.items()})"
done; done
