#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# End-to-end calibration lines (bench.py --phase) + their rocprofv3 kernel tables; run on the GPU box from the repo root.
#   tools/calib_lines.sh r3  -> gpurun_out/calib/: r3_calib_naive_line.json, r3_calib_kl_line.json, *_kernel_stats_top30.csv
set -u
TAG=${1:-r3}
R=$(pwd); OUT=$R/gpurun_out/calib; rm -rf $OUT; mkdir -p $OUT
NAIVE="--phase calib-naive --model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --steps 60 --warmup 5"
KL="--phase calib-kl --model resnet50_v1 --quant-type channel --steps 12 --warmup 2"
python3 bench.py $NAIVE > $OUT/${TAG}_calib_naive_line.json 2> $OUT/naive.err
python3 bench.py $KL > $OUT/${TAG}_calib_kl_line.json 2> $OUT/kl.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_naive -o bench -- python3 $R/bench.py $NAIVE --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/t_naive.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_kl -o bench -- python3 $R/bench.py $KL --no-cpu-baseline --no-kernel-events > /dev/null 2> $OUT/t_kl.err
cd $R
head -31 $(find $OUT/t_naive -name '*kernel_stats.csv' | head -1) | cut -c1-200 > $OUT/${TAG}_calib_naive_kernel_stats_top30.csv
head -31 $(find $OUT/t_kl -name '*kernel_stats.csv' | head -1) | cut -c1-200 > $OUT/${TAG}_calib_kl_kernel_stats_top30.csv
rm -rf $OUT/t_naive $OUT/t_kl
python3 - $OUT/${TAG}_calib_naive_line.json $OUT/${TAG}_calib_kl_line.json <<'P'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    k = d["roofline"]["kernels"]
    print("%s\n  %.1f img/s  %.3f ms/step  repeats %d | %s" % (d["metric"], d["value"], d["ms_per_step"], d["repeats"],
          "  ".join("%s %.3f ms (%.2f)" % (n, k[n]["ms_per_step"], k[n]["frac"]) for n in sorted(k, key=lambda n: -k[n]["ms_per_step"]))))
    print("  cpu_baseline:", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("sample"))
    if "kl_search" in d: print("  kl_search:", d["kl_search"])
P
