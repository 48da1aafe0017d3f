#!/usr/bin/env python3
"""Alternating A/B of library / environment variants on bench.py inside ONE GPU call.  A box of this pool wanders by +-3 %
between consecutive runs, so single runs say nothing: every round runs every variant once, in order; reported per variant are
mean / median / min / max images/s and the mean of the PAIRED ratios against the first variant (same round = same mood).

    python tools/ab.py [--rounds 5] [--args "bench.py args"] "label" "label|ENV=V ENV2=V" "label||--streams 2" ...
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    rounds, extra = 5, "--steps 300 --warmup 10 --no-cpu-baseline --no-headline --no-kernel-events"
    while argv and argv[0].startswith("--"):
        if argv[0] == "--rounds":
            rounds = int(argv[1])
        elif argv[0] == "--args":
            extra = extra + " " + argv[1]
        argv = argv[2:]
    variants = []
    for a in argv:
        parts = a.split("|")
        label, envs, vargs = parts[0], (parts[1] if len(parts) > 1 else ""), (parts[2] if len(parts) > 2 else "")
        variants.append((label, dict(kv.split("=", 1) for kv in envs.split()) if envs else {}, vargs.split()))
    vals = {label: [] for label, _, _ in variants}
    base_env = dict(os.environ)
    base_env.setdefault("FQ_BENCH_MIN_REGION_S", "3")
    for r in range(rounds):
        for label, env, vargs in variants:
            res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra.split() + vargs, env=dict(base_env, **env),
                                 capture_output=True, text=True, cwd=ROOT)
            try:
                v = json.loads(res.stdout.strip().splitlines()[-1])["value"]
            except Exception:
                v = float("nan")
            vals[label].append(v)
            print("round %d  %-32s %10.1f img/s" % (r + 1, label, v), flush=True)
    first = variants[0][0]
    print("%-32s %10s %10s %10s %10s   paired vs %s" % ("variant", "mean", "median", "min", "max", first))
    for label, _, _ in variants:
        x = [v for v in vals[label] if v == v]
        ratios = [a / b for a, b in zip(vals[label], vals[first]) if a == a and b == b]
        if not x or not ratios:
            print("%-32s (no run of this variant produced a line)" % label)
            continue
        print("%-32s %10.1f %10.1f %10.1f %10.1f   %+.2f %% (sd %.2f)" % (
            label, statistics.mean(x), statistics.median(x), min(x), max(x), (statistics.mean(ratios) - 1) * 100,
            statistics.pstdev(ratios) * 100 if len(ratios) > 1 else 0.0))


if __name__ == "__main__":
    main()
