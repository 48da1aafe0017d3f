#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# Round-6 evidence in one GPU call (run from the repo root on the GPU box): everything tools/refresh_r5.sh collects (default
# workload: rocprofv3 tables, PMC traffic, events-vs-rocprof check, the driver's line; the other BASELINE configurations with
# PMC traffic for the offline ones; calibration phases; CLI vs bench), then round 6's own: the recompute pairs alone (kernel
# durations, SQ counters, PMC traffic of the two new kernels), the A/B of the pairs in the step at three batch sizes, the
# batch / unseen-net sweep, the sliced-filter effect, eight ranks on one GPU.  Copy what should be judged into profiles/.
set -u
TAG=${1:-r6}
R=$(pwd); O=$R/gpurun_out/refresh
bash tools/refresh_r5.sh $TAG > gpurun_out/refresh_r5part.log 2>&1
# ---- the recompute pairs alone ---------------------------------------------------------------------------------------------
bash tools/pwdw_prof.sh 1,2,3,4,5 > $O/${TAG}_pwdw_kernel_times.txt 2>&1
bash tools/pwdw_pmc.sh 1,2,3 > /dev/null 2>&1; cp gpurun_out/pwdw_pmc/summary.txt $O/${TAG}_pwdw_pmc_sq.txt 2>/dev/null
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_pf -o b -- python3 $R/tools/pwdwbench.py --pairs 1,2,3 --reps 5 > /dev/null 2> $O/pmc_pf.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_pw -o b -- python3 $R/tools/pwdwbench.py --pairs 1,2,3 --reps 5 > /dev/null 2> $O/pmc_pw.err )
python3 - $O > $O/${TAG}_pwdw_pmc_traffic.txt <<'P'
import csv, glob, re, sys, collections
csv.field_size_limit(10 ** 9)
O = sys.argv[1]
def load(d, cname):
    agg = collections.defaultdict(list)
    for f in glob.glob(O + '/' + d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if r['Counter_Name'] == cname and ('pwdw_kernel' in n or 'pw_stat_kernel' in n or 'pwconv_stream_kernel' in n or 'dwconv3x3_cols4' in n):
                agg[re.sub(r'\(.*', '', n.replace('void (anonymous namespace)::', ''))[:44]].append(float(r['Counter_Value']))
    return agg
fe, wr = load('pmc_pf', 'FETCH_SIZE'), load('pmc_pw', 'WRITE_SIZE')
print("HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes around tools/pwdwbench.py --pairs 1,2,3, batch 128;")
print("KiB counters, FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes; MB = 1e6 bytes).  Algorithmic input / output of the pairs:")
print("pair 1: x 205.5 MB, y 411.0, z 102.8;  pair 2: x 102.8, y 205.5, z 205.5;  pair 3: x 205.5, y 205.5, z 51.4")
for k in sorted(set(fe) | set(wr)):
    f = sum(fe.get(k, [0])) / max(len(fe.get(k, [1])), 1) * 1024 * 2 / 1e6
    w = sum(wr.get(k, [0])) / max(len(wr.get(k, [1])), 1) * 1024 / 1e6
    print("  %-46s launches %3d  read %7.1f MB  written %7.1f MB" % (k, len(fe.get(k, [])), f, w))
P
rm -rf $O/pmc_pf $O/pmc_pw
# ---- the pairs in the step, three batch sizes ------------------------------------------------------------------------------
for b in 32 64 128; do
  python3 tools/ab.py --rounds 3 --args "--batch-size $b --steps 400" "two launches b$b|FQ_RECOMPUTE=0" "recompute pairs b$b|FQ_RECOMPUTE=1" 2>&1 | tail -3
done > $O/${TAG}_recompute_ab.txt
# ---- sweep, sliced filter, eight ranks --------------------------------------------------------------------------------------
bash tools/batch_sweep.sh $TAG > gpurun_out/refresh_sweep.log 2>&1
cp gpurun_out/sweep/${TAG}_batch_sweep.txt gpurun_out/sweep/${TAG}_smi.txt $O/ 2>/dev/null
python3 tools/sliced_effect.py 2>&1 | grep -v amdgpu > $O/${TAG}_sliced_effect.txt
for ph in eval calib-naive calib-kl; do
  FQ_BENCH_SHARE_GPU=1 FQ_BENCH_BACKEND=gloo FQ_DIST_BACKEND=gloo FQ_DIST_SHARE_GPU=1 python3 bench.py --gpus 8 --phase $ph --batch-size 16 --steps 10 --warmup 2 --no-cpu-baseline --no-headline --max-repeats 3 2>> $O/ranks8.err
done > $O/${TAG}_eight_ranks_one_gpu.jsonl
for ph in calib-naive; do
  python3 bench.py --gpus 1 --phase $ph --batch-size 16 --steps 10 --warmup 2 --no-cpu-baseline --no-headline --max-repeats 3 2>> $O/ranks8.err
done > $O/${TAG}_one_rank_same_batch.jsonl
ls -la $O | head -90
