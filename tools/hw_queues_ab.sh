export FQ_SYNTH_TRAIN_PER_CLASS=1 FQ_EVAL_TIMING=1 FQ_SYNTH_VAL_IMAGES=128000 FQ_BENCH_MIN_REGION_S=1
for q in 8 4 8 4; do
  for cfg in "mobilenet1.0|" "mobilenetv2_1.0|--quant-type channel --weight-bits 4 --offline" "resnet50_v1|--quant-type channel --offline"; do
    model=${cfg%%|*}; bextra=${cfg#*|}
    GPU_MAX_HW_QUEUES=$q python bench.py --model $model $bextra --steps 500 --no-cpu-baseline --no-headline --no-kernel-events 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH queues=$q $model: %.1f images/sec (%.4f ms/step)' % (l['value'], l['ms_per_step']))"
  done
done
A="--use-gpu 0 --pretrained false --synthetic-on-device"
for m in "mobilenet1.0|" "mobilenetv2_1.0|--quant-type channel --weight-bits-width 4 --quantize-input-offline --calib-epoch 1 --num-sample 1"; do
  model=${m%%|*}; extra=${m#*|}
  echo "== CLI $model (default flags)"; python examples/simulate_quantization.py --model $model $A $extra 2>/dev/null | grep "images/sec" | cut -c1-150
done
