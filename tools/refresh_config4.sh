export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
set -u
R=$(pwd); O=$R/gpurun_out/mini; rm -rf $O; mkdir -p $O
for cfg in "--model resnet50_v1 --quant-type channel" "--model resnet50_v1 --quant-type channel --offline" \
           "--model resnet50_v1 --quant-type channel --wino F43" "--model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline"; do
  python3 bench.py $cfg --steps 100 --no-cpu-baseline --no-headline >> $O/r4_other_configs.jsonl 2>> $O/other.err
done
for v in 1 0; do
  FQ_PWS_THIN=$v python3 bench.py --model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline --steps 200 --no-cpu-baseline --no-headline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']
print('FQ_PWS_THIN=$v mobilenetv2_1.0 offline:', l['value'], 'images/s', l['ms_per_step'], 'ms/step', {n:(round(v['ms_per_step'],3), v['frac']) for n,v in k.items()})" >> $O/r4_thin_ab.txt
done
BENCH_ARGS="--model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline --streams 1 --graph 0" MIN_US=8 bash tools/kprof.sh "" > $O/r4_kprof_mobilenetv2.txt 2>&1
python3 tools/dw16bench.py > $O/r4_dw16bench.txt 2>/dev/null
