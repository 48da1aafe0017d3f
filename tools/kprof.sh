#!/bin/bash
export FQ_BENCH_MIN_REGION_S=${FQ_BENCH_MIN_REGION_S:-1}   # (tools time with 1 s regions; the driver's plain bench.py run uses its 8 s default)
# Per-kernel averages of the benchmark step under rocprofv3 (kernel trace only), one table per variant, inside ONE GPU call:
#   tools/kprof.sh "" "FQ_PWS_AUTO=1 FQ_PWS_CFG=44"          (each argument: VAR=VALUE settings exported for that run)
# Dispatches are grouped by (kernel, grid size), so layers that share a kernel show up separately.  Prints calls per step,
# average us and us per step for every group above MIN_US (default 5) per step.  BENCH_ARGS adds bench.py flags; FILTER is a
# substring the kernel name must contain.
set -u
R=$(pwd); i=0
for v in "$@"; do
  i=$((i+1)); O=$R/gpurun_out/kprof_$i; rm -rf $O; mkdir -p $O
  ( cd /tmp && export TMPDIR=/tmp && export $v NOTHING=1 && rocprofv3 --kernel-trace --output-format csv -d $O -o bench -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-headline --no-kernel-events ${BENCH_ARGS:-} > $O/line.json 2> $O/err.txt )
  echo "== ${v:-(defaults)}   $(python3 -c "import json,sys; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print('%.0f img/s under rocprof' % d['value'])" 2>/dev/null)"
  python3 - $(find $O -name '*kernel_trace.csv' | head -1) "${FILTER:-}" "${MIN_US:-5}" <<'P'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt, min_us = sys.argv[2], float(sys.argv[3])
# steps the profiled process ran = launches of a kernel every step holds exactly once (bench.py repeats short blocks)
steps = float(sum(1 for r in rows if "stem_mfma" in r["Kernel_Name"] or "stem_conv3x3s2" in r["Kernel_Name"] or "stem7_pool" in r["Kernel_Name"] or "stem3_rows" in r["Kernel_Name"]) or 33)
g = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    g.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values()) / steps
for (name, grid), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    per_step = sum(v) / steps
    if per_step < min_us or flt not in name or len(v) < 0.9 * steps:     # (less than once per step: library warm-up / search)
        continue
    v.sort()
    print("  %-44s wgs %6d  calls/step %5.1f  median %7.1f us  per step %7.1f us" % (name[:44], grid, len(v) / steps, v[len(v) // 2], per_step))
print("  sum of all kernels per step: %.1f us" % tot)
P
  rm -rf $O
done
