#!/bin/bash
# does the CLI's evaluation (examples/simulate_quantization.py, the reference's entry point) reach bench.py's
# figure?  Synthetic ImageNet-shaped data generated on the device, 2000 batches of 128; --eval-graph 1 (default) vs 0.
# (run on the GPU box from the repo root; prints one line per run)
export FQ_SYNTH_VAL_IMAGES=${IMAGES:-256000} FQ_SYNTH_TRAIN_PER_CLASS=1
for cfg in "mobilenet1.0|" "mobilenetv2_1.0|--quant-type channel --weight-bits-width 4 --quantize-input-offline --calib-epoch 1 --num-sample 1"; do
  model=${cfg%%|*}; extra=${cfg#*|}
  for g in 1 0; do
    out=$(python examples/simulate_quantization.py --model $model --use-gpu 0 --pretrained false --synthetic-on-device --eval-graph $g $extra 2>/dev/null | grep "images/sec" | tail -1)
    echo "CLI   $model eval-graph=$g : $out"
  done
  bextra=""; [ "$model" = "mobilenetv2_1.0" ] && bextra="--quant-type channel --weight-bits 4 --offline"
  python bench.py --model $model $bextra --steps 1000 --no-cpu-baseline --no-headline --no-kernel-events 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH $model : %.1f images/sec (%.4f ms/step), single_stream %.1f' % (l['value'], l['ms_per_step'], (l.get('single_stream') or {}).get('value', 0)))"
done
