#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# Regenerates the rocprofv3 evidence under gpurun_out/refresh (run on the GPU box from the repo root); copy what should be
# judged into profiles/ afterwards (tools/refresh_profiles.sh r2 -> names prefixed r2_).  PMC passes are separate from each
# other and carry only --kernel-trace.
set -u
TAG=${1:-r2}
R=$(pwd); OUT=$R/gpurun_out/refresh; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-headline > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/trace.err
# (this one keeps the headline tensor: the table then holds absmax_per_sample_kernel + act_apply_kernel<true,...> on
# (128,64,112,112), the north star's kernel pair)
# the same steps one batch at a time and launched eagerly: with batches in flight a kernel's wall duration includes the time it
# shares the CUs with another batch's kernels (the durations of a step then add up to more than the step); this table is the
# one the event-timed per-kernel figures of the line (taken on steps that run alone) are to be compared with
# (FQ_BENCH_EVENT_BLOCK_EVERY=1, --event-every 10: bracketed steps in EVERY block of this short run, ten per block - what is
# checked here is the event method, not `value`; the driver's run keeps them in every third block)
FQ_BENCH_EVENT_BLOCK_EVERY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o bench -- python3 $R/bench.py --steps 100 --warmup 5 --streams 1 --graph 0 --event-every 10 --no-cpu-baseline > $OUT/${TAG}_bench_line_one_stream_under_rocprof.json 2> $OUT/trace1.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-headline --no-kernel-events > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-headline --no-kernel-events > /dev/null 2> $OUT/pmc_write.err
cd $R
F=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $F $W $OUT/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around bench.py --steps 5 --warmup 2, $TAG" > $OUT/${TAG}_pmc_bench.txt
# HBM traffic of the north-star kernel pair on the 411 MB headline tensor (separate passes, kernel trace only)
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_hfetch -o head -- python3 $R/tools/pmc_run.py > /dev/null 2> $OUT/pmc_hfetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_hwrite -o head -- python3 $R/tools/pmc_run.py > /dev/null 2> $OUT/pmc_hwrite.err
cd $R
HF=$(find $OUT/pmc_hfetch -name '*counter_collection.csv' | head -1); HW=$(find $OUT/pmc_hwrite -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $HF $HW /dev/null "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around tools/pmc_run.py (headline tensor 128x64x112x112), $TAG" > $OUT/${TAG}_pmc_headline.txt 2>&1
rm -rf $OUT/pmc_hfetch $OUT/pmc_hwrite
FQ_BENCH_MIN_REGION_S=8 python3 bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err   # (the line as the driver runs it)
python3 bench.py --streams 1 --graph 0 --no-cpu-baseline --no-headline > $OUT/${TAG}_bench_line_one_stream.json 2>> $OUT/bench.err
# (the table the line's per-kernel event figures agree with keeps the plain name)
cp $(find $OUT/trace1 -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats.csv
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats_default_in_flight.csv
# do the line's per-kernel figures follow from rocprofv3's table?  THE check: the line the profiled process printed against
# that process' own table (gated at 3 %); then the un-profiled lines against it (two processes of one box; the one-stream
# line gated as well: it is the un-profiled figures the driver's line carries)
python3 tools/check_events_vs_rocprof.py $OUT/${TAG}_bench_line_one_stream_under_rocprof.json $OUT/${TAG}_bench_kernel_stats.csv > $OUT/${TAG}_events_vs_rocprof.txt 2>&1
echo "same process (one stream, under rocprofv3): rc=$?" >> $OUT/${TAG}_events_vs_rocprof.txt
python3 tools/check_events_vs_rocprof.py $OUT/${TAG}_bench_line_one_stream.json $OUT/${TAG}_bench_kernel_stats.csv --cross-process-gate >> $OUT/${TAG}_events_vs_rocprof.txt 2>&1
echo "un-profiled one-stream line of the same box against that table (gated at 3 % on the dominant family): rc=$?" >> $OUT/${TAG}_events_vs_rocprof.txt
python3 tools/check_events_vs_rocprof.py $OUT/${TAG}_bench_line.json $OUT/${TAG}_bench_kernel_stats.csv >> $OUT/${TAG}_events_vs_rocprof.txt 2>&1
cat $OUT/${TAG}_events_vs_rocprof.txt
tail -c 600 $OUT/${TAG}_bench_line.json; cat $OUT/${TAG}_pmc_bench.txt; head -24 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-170
