#!/bin/bash
# The subsampled stage boundaries of the v1 bottleneck ResNets (fq_pwconv_i8_sub2 / fq_pwconv_i8_c16_dual_sub2) in the step:
# alternating A/Bs (tools/ab.py) of FQ_SUBSAMPLE=0 against the default, all in ONE GPU call; then one stream's per-kernel table.
#   bash tools/sub2_ab.sh      -> gpurun_out/sub2/ab.txt, gpurun_out/ktab/*_sub2.txt
set -u
O=gpurun_out/sub2; mkdir -p $O
ab() {  # label, rounds, bench arguments
  echo "## $1"
  python3 tools/ab.py --rounds $2 --args "$3" "whole trunk|FQ_SUBSAMPLE=0" "subsampled" 2>&1 | tail -3
}
{
  ab "resnet50_v1 per-channel W8A8 online, batch 128" 3 "--model resnet50_v1 --quant-type channel --steps 100"
  ab "resnet50_v1 per-channel W8A8 offline (BASELINE configuration 3), batch 128" 3 "--model resnet50_v1 --quant-type channel --offline --steps 100"
  ab "resnet50_v1 Winograd-domain F43 (BASELINE configuration 5), batch 128" 3 "--model resnet50_v1 --quant-type channel --wino F43 --steps 100"
  ab "resnet50_v1 online, batch 32" 2 "--model resnet50_v1 --quant-type channel --batch-size 32 --steps 200"
  ab "resnet50_v1 online, batch 64" 2 "--model resnet50_v1 --quant-type channel --batch-size 64 --steps 150"
  ab "resnet50_v1 online, batch 256" 2 "--model resnet50_v1 --quant-type channel --batch-size 256 --steps 50"
  ab "resnet101_v1 online, batch 128 (no rule was tuned on it)" 2 "--model resnet101_v1 --quant-type channel --steps 60"
  ab "resnet152_v1 offline, batch 128" 2 "--model resnet152_v1 --quant-type channel --offline --steps 40"
} > $O/ab.txt 2>&1
cat $O/ab.txt
bash tools/kernel_table.sh cfg3on_sub2 --model resnet50_v1 --quant-type channel > /dev/null 2>&1
bash tools/kernel_table.sh cfg3_sub2 --model resnet50_v1 --quant-type channel --offline > /dev/null 2>&1
