#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace --stats) of tools/pwdwbench.py: the two storing launches and the recompute pair
#   bash tools/pwdw_prof.sh [pairs] [extra pwdwbench args]    -> gpurun_out/pwdw_prof/stats.txt
set -u
PAIRS=${1:-1,2,3}
R=$(pwd); OUT=$R/gpurun_out/pwdw_prof; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o bench -- python3 $R/tools/pwdwbench.py --pairs $PAIRS --reps 20 > $OUT/bench.txt 2> $OUT/err.txt )
python3 - $OUT <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/t/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(sys.argv[1] + '/stats.txt', 'w') as out:
    for r in rows:
        n = r['Name']
        if '(anonymous namespace)' not in n:
            continue
        n = re.sub(r'\(.*', '', n.replace('void (anonymous namespace)::', ''))
        line = "%-52s calls %4s  avg %8.1f us  min %8.1f  max %8.1f" % (n[:52], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3)
        print(line); out.write(line + "\n")
PY
grep -v amdgpu.ids $OUT/bench.txt | tail -6
