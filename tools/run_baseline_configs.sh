set -u
export FQ_SYNTH_VAL_IMAGES=128 FQ_SYNTH_TRAIN_PER_CLASS=6
run() { echo "== $*"; timeout 500 python examples/simulate_quantization.py "$@" --use-gpu=0 2>&1 | grep -v amdgpu | grep -E "acc |avg_acc|speed|Error|error|Traceback" | tail -4; }
run --model=cifar_resnet20_v1 --dataset=cifar10 --batch-size 64
run --model=mobilenet1.0 --batch-size 64
FQ_SYNTH_TRAIN_PER_CLASS=1 run --model=resnet50_v1 --quant-type=channel --quantize-input-offline --calib-mode=kl --calib-epoch 1 --num-sample 1 --batch-size 32
FQ_SYNTH_TRAIN_PER_CLASS=1 run --model=mobilenetv2_1.0 --quant-type=channel --weight-bits-width 4 --quantize-input-offline --calib-mode=naive --calib-epoch 1 --num-sample 1 --batch-size 32
run --model=resnet50_v1 --quant-type=channel --wino_quantize=F43 --batch-size 32
