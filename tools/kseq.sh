#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# The kernels of ONE benchmark step in launch order, with the idle gap in front of each (rocprofv3 kernel trace):
#   tools/kseq.sh ["VAR=VALUE ..."]        BENCH_ARGS adds bench.py flags
# Prints the median step (by total span) of the timed steps: per kernel start-to-end us, gap to the previous kernel's end.
set -u
R=$(pwd); O=$R/gpurun_out/kseq; rm -rf $O; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && export ${1:-NOTHING=1} && rocprofv3 --kernel-trace --output-format csv -d $O -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-headline --no-kernel-events ${BENCH_ARGS:-} > $O/line.json 2> $O/err.txt )
python3 - $(find $O -name '*kernel_trace.csv' | head -1) <<'P'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
# a step starts at the stem kernel
starts = [i for i, r in enumerate(rows) if "stem" in nm(r)]
steps = [rows[a:b] for a, b in zip(starts, starts[1:])]
steps = [s for s in steps if len(s) == max(len(t) for t in steps[3:])] if len(steps) > 4 else steps
steps.sort(key=lambda s: int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]))
s = steps[len(steps) // 2]
t0 = int(s[0]["Start_Timestamp"]); prev = None; busy = 0; gaps = 0
for r in s:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (a - prev) / 1e3 if prev else 0.0
    print("  +%8.1f us  %-52s wgs %6d  %7.1f us   gap %5.1f" % ((a - t0) / 1e3, nm(r)[:52], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), (b - a) / 1e3, gap))
    busy += (b - a) / 1e3; gaps += max(gap, 0); prev = b
print("  %d kernels, first start to last end %.1f us: kernels %.1f us, gaps %.1f us" % (len(s), (prev - t0) / 1e3, busy, gaps))
P
rm -rf $O
