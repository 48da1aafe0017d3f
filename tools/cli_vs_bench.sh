#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# Does the CLI's evaluation (examples/simulate_quantization.py, the reference's entry point) reach bench.py's figure?  Synthetic
# ImageNet-shaped data generated on the device; --eval-graph 2 (the default: replay only where the host would hold the GPU back),
# 1 (always) and 0 (never) on a long pass (2000 batches of 128) and, for the default, on a 50 000-image pass (a validation set).
# Each run prints the whole-pass figure, the figure once set up, and what the set-up consisted of (FQ_EVAL_TIMING=1).
# (run on the GPU box from the repo root; prints one block per run)
export FQ_SYNTH_TRAIN_PER_CLASS=1 FQ_EVAL_TIMING=1
for cfg in "mobilenet1.0|" "mobilenetv2_1.0|--quant-type channel --weight-bits-width 4 --quantize-input-offline --calib-epoch 1 --num-sample 1"; do
  model=${cfg%%|*}; extra=${cfg#*|}
  for run in "256000 2" "256000 1" "256000 0" "50000 2"; do
    set -- $run
    out=$(FQ_SYNTH_VAL_IMAGES=$1 python examples/simulate_quantization.py --model $model --use-gpu 0 --pretrained false --synthetic-on-device --eval-graph $2 $extra 2>/dev/null | grep "images/sec\|^\[eval\] Eval" | tail -3)
    echo "CLI   $model, $1 images, eval-graph=$2 :"; echo "$out" | sed 's/^/      /'
  done
  bextra=""; [ "$model" = "mobilenetv2_1.0" ] && bextra="--quant-type channel --weight-bits 4 --offline"
  python bench.py --model $model $bextra --steps 1000 --no-cpu-baseline --no-headline --no-kernel-events 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH $model : %.1f images/sec (%.4f ms/step), single_stream %.1f' % (l['value'], l['ms_per_step'], (l.get('single_stream') or {}).get('value', 0)))"
done
