#!/bin/bash
# Robustness sweep: bench.py at batch 32 / 64 / 128 / 256 for BASELINE configurations 2-5 and at batch 128
# for two nets of the zoo no heuristic was tuned on (mobilenet0.5, resnet18_v1), all in ONE GPU call so that the rows share a
# box.  Also records the box's clocks / power cap before and after (item 6: which box is this?).  Run from the repo root:
#   bash tools/batch_sweep.sh [tag]     -> gpurun_out/sweep/<tag>_batch_sweep.{jsonl,txt}
set -u
TAG=${1:-r6}
R=$(pwd); O=$R/gpurun_out/sweep; mkdir -p $O
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-2}
J=$O/${TAG}_batch_sweep.jsonl; : > $J
smi() { ( rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showtemp 2>&1 | grep -vE '^=|^$|WARNING' | head -40 ) ; }
{ echo "## rocm-smi before"; smi; } > $O/${TAG}_smi.txt
line() {   # name, then bench.py arguments
  local name=$1; shift
  for b in $BATCHES; do
    local steps=$(( 12800 / b )); [ $steps -lt 40 ] && steps=40
    python3 bench.py "$@" --batch-size $b --steps $steps --warmup 10 --no-cpu-baseline --no-headline 2>> $O/sweep.err |
      python3 -c "import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    d['sweep'] = {'name': '$name', 'batch': $b}
    print(json.dumps(d))" >> $J
  done
}
BATCHES="32 64 128 256"
line "cfg2 mobilenet1.0 layer online"
line "cfg3 resnet50_v1 channel offline" --model resnet50_v1 --quant-type channel --offline
line "cfg3o resnet50_v1 channel online" --model resnet50_v1 --quant-type channel
line "cfg4 mobilenetv2_1.0 channel w4 offline" --model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline
line "cfg5 resnet50_v1 channel F43" --model resnet50_v1 --quant-type channel --wino F43
BATCHES="128"
line "mobilenet0.5 layer online" --model mobilenet0.5
line "mobilenet0.25 layer online" --model mobilenet0.25
line "resnet18_v1 channel online" --model resnet18_v1 --quant-type channel
line "resnet34_v1 channel online" --model resnet34_v1 --quant-type channel
{ echo "## rocm-smi after"; smi; } >> $O/${TAG}_smi.txt
python3 tools/sweep_table.py $J > $O/${TAG}_batch_sweep.txt
cat $O/${TAG}_batch_sweep.txt
