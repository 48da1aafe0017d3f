#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# Round evidence in one GPU call (run from the repo root on the GPU box): default-workload rocprof trace + PMC traffic +
# bench line (tools/refresh_profiles.sh), calibration kernels, the other BASELINE configurations (bench line + rocprofv3
# kernel table each), SQ counters, host baseline.  Everything lands in gpurun_out/refresh*/; copy into profiles/.
set -u
TAG=${1:-r2}
R=$(pwd)
bash tools/refresh_profiles.sh $TAG > gpurun_out/refresh_main.log 2>&1
O=$R/gpurun_out/refresh
python3 tools/calibbench.py --json $O/${TAG}_calibbench.json > $O/${TAG}_calibbench.txt 2>&1
python3 tools/hostbench.py --batch 128 --reps 3 > $O/${TAG}_hostbench.json 2> /dev/null
i=0
for cfg in "--model resnet50_v1 --quant-type channel" "--model resnet50_v1 --quant-type channel --offline" \
           "--model resnet50_v1 --quant-type channel --wino F43" "--model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline"; do
  i=$((i+1))
  python3 bench.py $cfg --steps 100 --no-cpu-baseline --no-headline >> $O/${TAG}_other_configs.jsonl 2>> $O/other.err
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg$i -o bench -- python3 $R/bench.py $cfg --steps 30 --warmup 6 --no-cpu-baseline --no-headline --no-kernel-events > /dev/null 2>> $O/other.err )
  cp $(find $O/trace_cfg$i -name '*kernel_stats.csv' | head -1) $O/${TAG}_cfg${i}_kernel_stats.csv
  rm -rf $O/trace_cfg$i
done
bash tools/pmc_sq.sh > $O/${TAG}_pmc_sq.log 2>&1
cp gpurun_out/pmc_sq/summary.txt $O/${TAG}_pmc_sq.txt 2>/dev/null
rm -rf $O/trace $O/pmc_fetch $O/pmc_write gpurun_out/pmc_sq
ls -la $O | head -40
