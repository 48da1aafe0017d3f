#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}
# The parts of tools/refresh_r6.sh that the ResNet changes of round 6 (subsampled stage boundaries, residual operand requested ahead)
# made stale - the default workload's files are untouched by them: PMC traffic of configuration 3, the lines and kernel tables of
# the other BASELINE configurations, the batch / unseen-net sweep.  One GPU call; copy what should be judged into profiles/.
set -u
TAG=${1:-r6}
R=$(pwd); O=$R/gpurun_out/refresh; mkdir -p $O
rm -f $O/${TAG}_other_configs.jsonl
for cfg in "resnet50_v1_channel_w8a8_offline|--model resnet50_v1 --quant-type channel --offline"; do
  key=${cfg%%|*}; args=${cfg#*|}
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$key -o bench -- python3 $R/bench.py $args --steps 3 --warmup 2 --min-region-s 0 --no-cpu-baseline --no-headline --no-kernel-events > /dev/null 2> $O/pmc_f_$key.err
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$key -o bench -- python3 $R/bench.py $args --steps 3 --warmup 2 --min-region-s 0 --no-cpu-baseline --no-headline --no-kernel-events > /dev/null 2> $O/pmc_w_$key.err )
  F=$(find $O/pmc_f_$key -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_w_$key -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py $F $W $O/pmc_traffic_$key.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around bench.py $args --steps 3 --warmup 2, $TAG" > $O/${TAG}_pmc_$key.txt 2>&1
  cp $O/pmc_traffic_$key.json profiles/pmc_traffic_$key.json
  rm -rf $O/pmc_f_$key $O/pmc_w_$key
done
i=0
for cfg in "--model resnet50_v1 --quant-type channel" "--model resnet50_v1 --quant-type channel --offline" \
           "--model resnet50_v1 --quant-type channel --wino F43" "--model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline"; do
  i=$((i+1))
  python3 bench.py $cfg --steps 100 --no-cpu-baseline --no-headline >> $O/${TAG}_other_configs.jsonl 2>> $O/other.err
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg$i -o bench -- python3 $R/bench.py $cfg --steps 30 --warmup 6 --no-cpu-baseline --no-headline --no-kernel-events > /dev/null 2>> $O/other.err )
  head -31 $(find $O/trace_cfg$i -name '*kernel_stats.csv' | head -1) | cut -c1-220 > $O/${TAG}_cfg${i}_kernel_stats_top30.csv
  rm -rf $O/trace_cfg$i
done
python3 - $O/${TAG}_other_configs.jsonl > $O/${TAG}_other_configs_summary.txt <<'P'
import json, sys
for ln in open(sys.argv[1]):
    try:
        d = json.loads(ln)
    except Exception:
        continue
    r = d["roofline"]; k = r["kernels"]; t = r.get("traffic_from_profiles")
    print("%s\n  %.1f img/s %.4f ms/step | whole step frac %.3f (4 B per element: %.3f) | %s" % (
        d["config"]["workload"][:90], d["value"], d["ms_per_step"], r["whole_step"]["frac"], r["whole_step"]["frac_algorithmic"],
        "  ".join("%s %.3f ms frac %.2f (%.2f)" % (n, k[n]["ms_per_step"], k[n]["frac"], k[n]["frac_algorithmic"]) for n in sorted(k, key=lambda n: -k[n]["ms_per_step"]))))
    if t:
        print("  PMC traffic of %s: %.1f MB per launch (moved by the line's count: %.1f MB)" % (t["kernel"], t["hbm_bytes_per_launch"] / 1e6, k[t["kernel"]]["moved_bytes_per_launch"] / 1e6))
P
bash tools/batch_sweep.sh $TAG > gpurun_out/refresh_sweep.log 2>&1
cp gpurun_out/sweep/${TAG}_batch_sweep.txt gpurun_out/sweep/${TAG}_smi.txt $O/ 2>/dev/null
# the calibration lines of configuration 3 (KL collection) and the default line, for the record of the final tree
python3 bench.py --phase calib-kl --model resnet50_v1 --quant-type channel --steps 10 --warmup 2 --no-cpu-baseline --no-headline > $O/${TAG}_calib_kl_line.json 2>> $O/other.err
python3 bench.py > $O/${TAG}_bench_line_final.json 2>> $O/other.err
cat $O/${TAG}_other_configs_summary.txt; cat $O/${TAG}_batch_sweep.txt
