#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# ResNet-50 (online, and F43) with the closing 1x1 of the 56x56 units on the split form (FQ_PWS_RES_SPLIT=1) or the streaming form (0)
for r in 1 2 3; do for m in 1 0; do
  FQ_PWS_RES_SPLIT=$m python3 bench.py --model resnet50_v1 --quant-type channel --steps 150 --no-cpu-baseline --no-headline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']
print('FQ_PWS_RES_SPLIT=$m resnet50 online:', l['value'], 'images/s; one batch at a time', l['single_stream']['value'], {n:round(v['ms_per_step'],3) for n,v in k.items()})"
done; done
