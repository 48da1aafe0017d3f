#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ (run on the GPU box from the repo root); copy what should be
# judged into profiles/ afterwards.  PMC passes are separate from each other and carry only --kernel-trace.
set -u
R=$(pwd); OUT=$R/gpurun_out/refresh; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/line_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/pmc_write.err
cd $R
F=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $F $W $OUT/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around bench.py --steps 5 --warmup 2" > $OUT/pmc_bench.txt
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench.err
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/trace -name '*domain_stats.csv' | head -1) $OUT/domain_stats.csv 2>/dev/null
tail -c 600 $OUT/bench_line.json; cat $OUT/pmc_bench.txt; head -12 $OUT/kernel_stats.csv | cut -c1-150
