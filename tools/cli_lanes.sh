export FQ_SYNTH_TRAIN_PER_CLASS=1 FQ_EVAL_TIMING=1 FQ_SYNTH_VAL_IMAGES=128000 FQ_BENCH_MIN_REGION_S=1
for cfg in "mobilenetv2_1.0|--quant-type channel --weight-bits-width 4 --quantize-input-offline --calib-epoch 1 --num-sample 1|--quant-type channel --weight-bits 4 --offline" "mobilenet1.0||"; do
  model=${cfg%%|*}; rest=${cfg#*|}; extra=${rest%%|*}; bextra=${rest#*|}
  for s in 2 3 4 3 4; do
    echo "== CLI $model lanes $s"; python examples/simulate_quantization.py --model $model --use-gpu 0 --pretrained false --synthetic-on-device --synthetic-resident 12 --eval-graph 1 --eval-streams $s $extra 2>/dev/null | grep "^\[eval\] Eval" | sed 's/set-up.*replayed,//' | cut -c1-200
    python examples/simulate_quantization.py --model $model --use-gpu 0 --pretrained false --synthetic-on-device --synthetic-resident 12 --eval-graph 1 --eval-streams $s $extra 2>/dev/null | grep "once set up" | cut -c1-60
  done
  for s in 2 3 4 3 4; do
    python bench.py --model $model $bextra --steps 1000 --streams $s --no-cpu-baseline --no-headline --no-kernel-events 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BENCH $model streams $s: %.1f images/sec (%.4f ms/step)' % (l['value'], l['ms_per_step']))"
  done
done
