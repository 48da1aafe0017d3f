#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# nn.Conv2D(quantized=True): GPU tests + the quantized_mobilenet bench in its three launch modes + the default bench line
# (run on the GPU box from the repo root; results under gpurun_out/)
TAG=${1:-r4c}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_qconv.py tests/test_gpu_qconv_net.py -q -m gpu -x > gpurun_out/${TAG}_qconv.log 2>&1; tail -30 gpurun_out/${TAG}_qconv.log
timeout 600 python -m pytest tests/test_gpu_net.py -q -m gpu -x -k "quantized or calibration_on" > gpurun_out/${TAG}_net.log 2>&1; tail -5 gpurun_out/${TAG}_net.log
for a in "" "--no-fuse" "--streams 1 --graph 0" "--streams 1 --graph 0 --no-fuse"; do
  python bench.py --model quantized_mobilenet1.0 --steps 100 --no-cpu-baseline --no-headline $a > gpurun_out/${TAG}_qbench.json 2> gpurun_out/${TAG}_qbench.err
  python - "$a" gpurun_out/${TAG}_qbench.json gpurun_out/${TAG}_qbench.err <<'P'
import json, sys
a, path, err = sys.argv[1:4]
try:
    l = json.loads(open(path).read().strip().splitlines()[-1])
    r = l["roofline"]
    print("ARGS [%s]" % a, l["value"], l["ms_per_step"], (l.get("single_stream") or {}).get("value"), (r["kernel"] or "")[:30], r["frac"])
    for k, v in r["kernels"].items():
        print("    %-10s %8.2f us/launch  frac %.3f  launches %d  ms/step %.4f" % (k, v["avg_launch_us"], v["frac"], v["launches"], v["ms_per_step"]))
except Exception as e:
    print("ARGS [%s] FAILED" % a, e)
    print(open(err).read()[-2500:])
P
done
python bench.py --steps 200 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2>/dev/null
python - gpurun_out/${TAG}_bench.json <<'P'
import json, sys
l = json.load(open(sys.argv[1])); r = l["roofline"]
print("default:", l["value"], l["ms_per_step"], r["frac"], r["frac_raw_events"], r["launch_overhead_us_removed"], r["event_pair_minus_null_kernel_us"])
P
