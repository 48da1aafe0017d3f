#!/usr/bin/env python3
"""How many kernels are in flight, and how long each takes there: from a rocprofv3 --kernel-trace csv of a bench.py run with
several batches in flight.  Over the last `--window-ms` of the trace (the timed region): the share of time with 0, 1, 2 ...
kernels running, the mean number running, and per kernel name the mean duration inside the window (compare with the one-stream
table of tools/step_table.py: a kernel that takes 3x as long among three others got a third of the GPU).
    python tools/lane_overlap.py <kernel_trace.csv> [--window-ms 200]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:58]


def main():
    path = sys.argv[1]
    win = float(sys.argv[sys.argv.index("--window-ms") + 1]) if "--window-ms" in sys.argv else 200.0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    t_end = max(e for _, e, _ in rows)
    t0 = t_end - int(win * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist = defaultdict(float)
    cur, last = 0, ev[0][0]
    for t, d in ev:
        hist[cur] += t - last
        last = t
        cur += d
    span = float(ev[-1][0] - ev[0][0])
    print("window %.1f ms, %d launches" % (span / 1e6, len(rows)))
    mean = sum(k * v for k, v in hist.items()) / span
    print("kernels in flight: " + "  ".join("%d: %.1f %%" % (k, 100.0 * hist[k] / span) for k in sorted(hist)) + "   mean %.2f" % mean)
    per = defaultdict(list)
    for s, e, n in rows:
        per[n].append((e - s) / 1e3)
    tot = sum(sum(v) for v in per.values())
    print("%-58s %6s %10s %8s" % ("kernel", "calls", "mean us", "share"))
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:24]:
        print("%-58s %6d %10.1f %7.1f%%" % (n, len(v), sum(v) / len(v), 100.0 * sum(v) / tot))


if __name__ == "__main__":
    main()
