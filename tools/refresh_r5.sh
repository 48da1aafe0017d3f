#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# Round-5 evidence in one GPU call (run from the repo root on the GPU box).  Everything tools/refresh_all.sh collects for the
# default workload (rocprofv3 tables, PMC traffic incl. the headline tensor, events-vs-rocprof check, SQ counters, calibration
# kernels, host baseline), then: PMC FETCH / WRITE passes for BASELINE configurations 3 and 4 under offline thresholds (written
# as profiles-ready pmc_traffic_<key>.json BEFORE their bench lines are taken, so that the lines quote them), the other
# configurations' lines with frac_actual, the calibration phases, the nn.Conv2D net, CLI vs bench.py, the first convolutions,
# the pointwise forms.  Copy what should be judged from gpurun_out/refresh*/ into profiles/.
set -u
TAG=${1:-r5}
R=$(pwd); O=$R/gpurun_out/refresh
bash tools/refresh_profiles.sh $TAG > gpurun_out/refresh_main.log 2>&1
python3 tools/calibbench.py --json $O/${TAG}_calibbench.json > $O/${TAG}_calibbench.txt 2>&1
python3 tools/hostbench.py --batch 128 --reps 3 > $O/${TAG}_hostbench.json 2> /dev/null
# ---- PMC traffic of configurations 3 and 4 (offline thresholds: the code-hand-over runs) ----------------------------------
for cfg in "resnet50_v1_channel_w8a8_offline|--model resnet50_v1 --quant-type channel --offline" \
           "mobilenetv2_1.0_channel_w4a8_offline|--model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline"; do
  key=${cfg%%|*}; args=${cfg#*|}
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$key -o bench -- python3 $R/bench.py $args --steps 3 --warmup 2 --min-region-s 0 --no-cpu-baseline --no-headline --no-kernel-events > /dev/null 2> $O/pmc_f_$key.err
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$key -o bench -- python3 $R/bench.py $args --steps 3 --warmup 2 --min-region-s 0 --no-cpu-baseline --no-headline --no-kernel-events > /dev/null 2> $O/pmc_w_$key.err )
  F=$(find $O/pmc_f_$key -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_w_$key -name '*counter_collection.csv' | head -1)
  python3 tools/pmc_summary.py $F $W $O/pmc_traffic_$key.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around bench.py $args --steps 3 --warmup 2, $TAG" > $O/${TAG}_pmc_$key.txt 2>&1
  cp $O/pmc_traffic_$key.json profiles/pmc_traffic_$key.json      # (the lines below quote it as traffic_from_profiles)
  rm -rf $O/pmc_f_$key $O/pmc_w_$key
done
# ---- the other BASELINE configurations: bench line + rocprofv3 kernel table each --------------------------------------------
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
bash tools/pmc_sq.sh > $O/${TAG}_pmc_sq.log 2>&1
cp gpurun_out/pmc_sq/summary.txt $O/${TAG}_pmc_sq.txt 2>/dev/null
bash tools/calib_lines.sh $TAG > gpurun_out/refresh_calib.log 2>&1
cp gpurun_out/calib/${TAG}_* $O/ 2>/dev/null
# ---- nn.Conv2D(quantized=True) MobileNet, CLI, first convolutions, pointwise forms ----------------------------------------------
python3 bench.py --model quantized_mobilenet1.0 > $O/${TAG}_qconv_line.json 2> $O/qconv.err
bash tools/cli_vs_bench.sh > $O/${TAG}_cli_vs_bench.txt 2>&1
python3 tools/stembench.py > $O/${TAG}_stembench.txt 2>/dev/null
( python3 tools/pwforms.py; python3 tools/pwforms.py --resnet ) > $O/${TAG}_pwforms.txt 2>/dev/null
rm -rf $O/trace $O/trace1 $O/pmc_fetch $O/pmc_write gpurun_out/pmc_sq
ls -la $O | head -80
