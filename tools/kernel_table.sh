#!/bin/bash
# Per-kernel time of one bench.py configuration on ONE stream (rocprofv3 --kernel-trace --stats): tools/kernel_table.sh <tag> <bench args...>
set -u
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/ktab; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/$TAG -o bench -- python3 $R/bench.py "$@" --steps 40 --warmup 5 --streams 1 --graph 0 --no-cpu-baseline --no-headline --no-kernel-events --min-region-s 0 --max-repeats 1 > $O/$TAG.json 2> $O/$TAG.err )
python3 - $O/$TAG $O/$TAG.txt <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(sys.argv[2], 'w') as out:
    for r in rows[:80]:
        n = r['Name']
        n = n.replace('(anonymous namespace)::', '').replace('void ', '')
        n = re.sub(r'\((?:[^()]|\([^()]*\))*\)\s*(\[clone.*)?$', '', n)    # the argument list only
        line = "%-76s calls %5s  avg %8.1f us  total %9.1f us  %5s %%" % (n[:76], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3, r['Percentage'][:5])
        out.write(line + "\n")
PY
rm -rf $O/$TAG
head -30 $O/$TAG.txt
