#!/bin/bash
# The folded shortcut (fq_pwconv_i8_shortcut) in the step: alternating A/Bs (tools/ab.py) of FQ_SHORTCUT_FUSE=0 against the default, ONE GPU call.
#   bash tools/shortcut_ab.sh      -> gpurun_out/shortcut/ab.txt, gpurun_out/ktab/cfg3on_short.txt
set -u
O=gpurun_out/shortcut; mkdir -p $O
ab() { echo "## $1"; python3 tools/ab.py --rounds $2 --args "$3" "two launches|FQ_SHORTCUT_FUSE=0" "folded shortcut" 2>&1 | tail -3; }
{
  ab "resnet50_v1 per-channel W8A8 online, batch 128" 3 "--model resnet50_v1 --quant-type channel --steps 100"
  ab "resnet50_v1 Winograd-domain F43 (BASELINE configuration 5), batch 128" 3 "--model resnet50_v1 --quant-type channel --wino F43 --steps 100"
  ab "resnet50_v1 online, batch 32" 2 "--model resnet50_v1 --quant-type channel --batch-size 32 --steps 200"
  ab "resnet50_v1 online, batch 64" 2 "--model resnet50_v1 --quant-type channel --batch-size 64 --steps 150"
  ab "resnet101_v1 online, batch 128 (no rule was tuned on it)" 2 "--model resnet101_v1 --quant-type channel --steps 60"
  ab "resnet50_v1 per-channel W8A8 offline (BASELINE configuration 3), batch 128" 3 "--model resnet50_v1 --quant-type channel --offline --steps 100"
  ab "resnet152_v1 offline, batch 128" 2 "--model resnet152_v1 --quant-type channel --offline --steps 40"
} > $O/ab.txt 2>&1
cat $O/ab.txt
bash tools/kernel_table.sh cfg3on_short --model resnet50_v1 --quant-type channel > /dev/null 2>&1
head -28 gpurun_out/ktab/cfg3on_short.txt | cut -c1-150
