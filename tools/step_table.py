#!/usr/bin/env python3
"""Per-launch table of ONE step from a rocprofv3 --kernel-trace csv (…_kernel_trace.csv): the trace is cut into steps at each
launch of the stem kernel, steps with the most common launch count are kept, and for every position in the step the median
duration over those steps is printed beside the kernel's name and grid.  For a one-stream eager run (bench.py --streams 1
--graph 0) the column adds up to the step: this is the table that says which LAYER the time is in, which the per-family
figures of the bench line do not.
    python tools/step_table.py gpurun_out/x/…_kernel_trace.csv [first-kernel-substring]"""
import csv
import re
import statistics
import sys
from collections import Counter


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:58]


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "stem"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)))
    rows.sort()
    steps, cur = [], None
    for r in rows:
        if first in r[2]:
            if cur:
                steps.append(cur)
            cur = []
        if cur is not None:
            cur.append(r)
    if cur:
        steps.append(cur)
    sig = Counter(tuple(short(r[2]) for r in s) for s in steps).most_common(1)[0][0]
    keep = [s for s in steps if tuple(short(r[2]) for r in s) == sig]
    print("%d steps cut at '%s', %d with the most common launch sequence (%d launches)" % (len(steps), first, len(keep), len(sig)))
    tot = 0.0
    for i, name in enumerate(sig):
        d = statistics.median((s[i][1] - s[i][0]) / 1e3 for s in keep)
        gap = statistics.median((s[i][0] - s[i - 1][1]) / 1e3 for s in keep) if i else 0.0
        tot += d
        print("%3d %-58s wgs %6d x %4d  %8.1f us  (gap before %5.1f)" % (i, name, keep[0][i][3] // max(keep[0][i][4], 1), keep[0][i][4], d, gap))
    span = statistics.median((s[-1][1] - s[0][0]) / 1e3 for s in keep)
    print("sum of kernels %.1f us, first start -> last end %.1f us" % (tot, span))


if __name__ == "__main__":
    main()
