#!/usr/bin/env python3
"""Vector-unit work per kernel of one step, from `rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace` around a one-stream eager
bench.py run.  The dispatches are cut into steps at each launch of the stem kernel and the steps with the most common launch
sequence are kept (set-up, calibration and warm-up launch other kernels); per kernel name: launches per step, VALU wavefront
instructions per step and the time they need at one instruction per 4 cycles and SIMD on 1024 SIMDs at 2.4 GHz - the floor the
kernel (and, summed, the step) cannot go below however many batches are in flight.  (Packed fp32 instructions with three
distinct register-pair operands take two passes, so the floor is a lower bound.)
    python tools/valu_table.py <counter_collection.csv> [first-kernel-substring]"""
import csv
import re
import sys
from collections import Counter, defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:60]


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "stem"
    disp = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "SQ_INSTS_VALU":
            continue
        d = int(r["Dispatch_Id"])
        name, v = disp.get(d, (short(r["Kernel_Name"]), 0.0))
        disp[d] = (name, v + float(r["Counter_Value"]))
    seq = [disp[d] for d in sorted(disp)]
    steps, cur = [], None
    for name, v in seq:
        if first in name:
            if cur:
                steps.append(cur)
            cur = []
        if cur is not None:
            cur.append((name, v))
    if cur:
        steps.append(cur)
    sig = Counter(tuple(n for n, _ in s) for s in steps).most_common(1)[0][0]
    keep = [s for s in steps if tuple(n for n, _ in s) == sig]
    per = defaultdict(lambda: [0, 0.0])
    for i, name in enumerate(sig):
        per[name][0] += 1
        per[name][1] += sum(s[i][1] for s in keep) / len(keep)
    print("%d steps, %d with the most common launch sequence (%d launches)" % (len(steps), len(keep), len(sig)))
    print("%-60s %6s %14s %10s" % ("kernel", "calls", "VALU inst/step", "floor us"))
    tot = 0.0
    for name, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        us = v * 4.0 / 1024.0 / 2400.0
        tot += us
        print("%-60s %6d %14.3e %10.1f" % (name, n, v, us))
    print("sum of floors: %.1f us per step" % tot)


if __name__ == "__main__":
    main()
