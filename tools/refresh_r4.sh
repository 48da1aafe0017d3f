#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# Round-4 evidence in one GPU call (run from the repo root on the GPU box): everything tools/refresh_all.sh collects (default
# workload: rocprofv3 tables with the headline-tensor kernels, PMC traffic incl. the headline tensor, events-vs-rocprof check,
# other BASELINE configurations, SQ counters, calibration kernels, host baseline), the calibration phases end to end, and the
# round's additions: the nn.Conv2D(quantized=True) MobileNet (bench line with cpu_baseline, its --no-fuse line, a one-stream
# rocprofv3 kernel table, the epilogue ablation), CLI vs bench.py, MobileNetV2 with and without the thin streaming form.
# Copy what should be judged from gpurun_out/refresh*/ into profiles/.
set -u
TAG=${1:-r4}
R=$(pwd); O=$R/gpurun_out/refresh
bash tools/refresh_all.sh $TAG > gpurun_out/refresh_all.log 2>&1
bash tools/calib_lines.sh $TAG > gpurun_out/refresh_calib.log 2>&1
python3 bench.py --model quantized_mobilenet1.0 > $O/${TAG}_qconv_line.json 2> $O/qconv.err
python3 bench.py --model quantized_mobilenet1.0 --no-fuse --no-cpu-baseline --no-headline > $O/${TAG}_qconv_nofuse_line.json 2>> $O/qconv.err
python3 bench.py --model quantized_mobilenet1.0 --streams 1 --graph 0 --no-cpu-baseline --no-headline > $O/${TAG}_qconv_line_one_stream.json 2>> $O/qconv.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_q -o bench -- python3 $R/bench.py --model quantized_mobilenet1.0 --steps 100 --warmup 5 --streams 1 --graph 0 --no-cpu-baseline --no-headline > $O/${TAG}_qconv_line_one_stream_under_rocprof.json 2>> $O/qconv.err )
cp $(find $O/trace_q -name '*kernel_stats.csv' | head -1) $O/${TAG}_qconv_kernel_stats.csv; rm -rf $O/trace_q
python3 tools/check_events_vs_rocprof.py $O/${TAG}_qconv_line_one_stream_under_rocprof.json $O/${TAG}_qconv_kernel_stats.csv --steps-from stem_mfma > $O/${TAG}_qconv_events_vs_rocprof.txt 2>&1
python3 tools/check_events_vs_rocprof.py $O/${TAG}_qconv_line_one_stream.json $O/${TAG}_qconv_kernel_stats.csv --steps-from stem_mfma >> $O/${TAG}_qconv_events_vs_rocprof.txt 2>&1
python3 tools/qconv_ablate.py > $O/${TAG}_qconv_ablate.txt 2>&1
bash tools/cli_vs_bench.sh > $O/${TAG}_cli_vs_bench.txt 2>&1
for v in 1 0; do
  FQ_PWS_THIN=$v python3 bench.py --model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline --steps 200 --no-cpu-baseline --no-headline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']
print('FQ_PWS_THIN=$v mobilenetv2_1.0 offline:', l['value'], 'images/s', l['ms_per_step'], 'ms/step', {n:(round(v['ms_per_step'],3), v['frac']) for n,v in k.items()})" >> $O/${TAG}_thin_ab.txt
done
# KL collection (config 3): the producers bin what they store (default) against one histogram pass per block (round 3)
for f in 1 0; do
  FQ_KL_FUSED_HIST=$f python3 bench.py --phase calib-kl --model resnet50_v1 --quant-type channel --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['roofline']['kernels']
print('FQ_KL_FUSED_HIST=$f resnet50_v1 KL collection:', l['value'], 'images/s', l['ms_per_step'], 'ms/batch', {n:(round(v['ms_per_step'],3), v['frac']) for n,v in k.items()}, l['split'])" >> $O/${TAG}_kl_fused_ab.txt
done
# ResNet-50 offline with and without the trunk's code copy (fq_pwconv_i8_c16_dual), alternating; the closing 1x1 alone
for r in 1 2; do for m in 0 512; do
  FQ_SIDE_MAX_CIN=$m python3 bench.py --model resnet50_v1 --quant-type channel --offline --steps 200 --no-cpu-baseline --no-headline --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FQ_SIDE_MAX_CIN=$m resnet50_v1 offline:', l['value'], 'images/s', l['ms_per_step'], 'ms/step, one batch at a time', l['single_stream']['value'])" >> $O/${TAG}_side_codes_ab_run.txt
done; done
python3 tools/dualbench.py >> $O/${TAG}_side_codes_ab_run.txt 2>/dev/null
python3 tools/dw16bench.py > $O/${TAG}_dw16bench.txt 2>/dev/null
python3 tools/c3bench.py > $O/${TAG}_c3bench.txt 2>/dev/null
python3 tools/stembench.py >> $O/${TAG}_c3bench.txt 2>/dev/null
python3 tools/cli_lane_probe.py 2>/dev/null | grep "^evaluate\|^mode\|^one lane\|^CLI" > $O/${TAG}_cli_lane_probe.txt
# per-layer state of the other steps (one stream, eager), the 3x3 kernel's phase stamps, the barrier probe
BENCH_ARGS="--model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline --streams 1 --graph 0" MIN_US=8 bash tools/kprof.sh "" > $O/${TAG}_kprof_mobilenetv2.txt 2>&1
BENCH_ARGS="--model resnet50_v1 --quant-type channel --streams 1 --graph 0" MIN_US=8 bash tools/kprof.sh "" > $O/${TAG}_kprof_resnet50.txt 2>&1
BENCH_ARGS="--model resnet50_v1 --quant-type channel --offline --streams 1 --graph 0" MIN_US=8 bash tools/kprof.sh "" > $O/${TAG}_kprof_resnet50_offline.txt 2>&1
[ -f build_tools/libfakequant_trace.so ] && python3 tools/c3_trace.py > $O/${TAG}_c3_trace.txt 2>/dev/null
[ -x build_tools/grid_barrier_probe ] && ./build_tools/grid_barrier_probe > $O/${TAG}_grid_barrier_probe.txt 2>&1
# the round's last A/Bs (each alternates its two settings inside one call) and the form sweeps
bash tools/thin_link_ab.sh > $O/${TAG}_unit_link_ab.txt 2>&1
bash tools/thin_tiles_ab.sh > $O/${TAG}_thin_tiles_ab.txt 2>&1
bash tools/thin_tiles_ab_r50.sh >> $O/${TAG}_thin_tiles_ab.txt 2>&1
bash tools/res_split_ab.sh > $O/${TAG}_res_split_ab.txt 2>&1
( python3 tools/pwforms.py; python3 tools/pwforms.py --resnet; python3 tools/pwforms.py --mobilenetv2 ) > $O/${TAG}_pwforms.txt 2>/dev/null
python3 tools/stembench.py > $O/${TAG}_stembench.txt 2>/dev/null
ls -la $O | head -80
