export FQ_SYNTH_VAL_IMAGES=128000 FQ_SYNTH_TRAIN_PER_CLASS=1
A="--model mobilenetv2_1.0 --use-gpu 0 --pretrained false --synthetic-on-device --quant-type channel --weight-bits-width 4 --quantize-input-offline --calib-epoch 1 --num-sample 1"
for v in 1 0; do echo "THIN=$v $(FQ_PWS_THIN=$v python examples/simulate_quantization.py $A 2>/dev/null | grep images/sec | tail -1)"; done
echo "eval-streams 1: $(python examples/simulate_quantization.py $A --eval-streams 1 2>/dev/null | grep images/sec | tail -1)"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cliprof -o cli -- python3 $GRAFT_REPO_ROOT/examples/simulate_quantization.py $A > /dev/null 2>&1
head -14 $(find /tmp/cliprof -name '*kernel_stats.csv' | head -1) | cut -c1-130,260-330
