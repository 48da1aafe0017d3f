export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
for r in 1 2 3; do for m in 4096 3000; do
  FQ_PWS_THIN_MIN_TILES=$m python3 bench.py --model resnet50_v1 --quant-type channel --offline --steps 150 --no-cpu-baseline --no-headline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']
print('FQ_PWS_THIN_MIN_TILES=$m resnet50 offline:', l['value'], 'images/s; one batch at a time', l['single_stream']['value'], {n:round(v['ms_per_step'],3) for n,v in k.items()})"
done; done
