#!/usr/bin/env python3
"""Kernel micro-benchmark for tuning (not the driver's bench.py): times the fake-quant entry points on the
MobileNet activation shapes with HIP events on torch's current stream and prints achieved algorithmic GB/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

SHAPES = [(128, 64, 112, 112), (128, 32, 112, 112), (128, 128, 56, 56), (128, 256, 28, 28), (128, 512, 14, 14),
          (128, 1024, 7, 7), (128, 1024)]


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only-headline", action="store_true")
    ap.add_argument("--policy-table", action="store_true", help="all shapes x a few policies, compact table")
    ap.add_argument("--policy-sweep", action="store_true",
                    help="re-run the headline shape in child processes under every FQ_POLICY_* combination")
    args = ap.parse_args()
    if args.policy_table:
        import subprocess
        import re
        table = {}
        for pol in ("default", 0, 3, 4, 7):
            env = dict(os.environ)
            if pol != "default":
                env.update(FQ_POLICY_ONLINE=str(pol), FQ_POLICY_OFFLINE=str(pol & 3))
            out = subprocess.run([sys.executable, __file__, "--iters", str(args.iters)], env=env,
                                 capture_output=True, text=True).stdout
            shape = None
            for l in out.splitlines():
                if l.startswith("shape"):
                    shape = l.split("  ")[0]
                m = re.match(r"\s+(\S+)\s+\(\d+B\)\s+median\s+([\d.]+) ms.*->\s+([\d.]+) GB/s", l)
                if m and shape:
                    table.setdefault((shape, m.group(1)), {})[pol] = float(m.group(3))
        print("%-28s %-10s %s" % ("shape", "case", "  ".join("%8s" % str(p) for p in ("default", 0, 3, 4, 7))))
        for (shape, case), row in table.items():
            print("%-28s %-10s %s" % (shape, case, "  ".join("%8.0f" % row.get(p, 0) for p in ("default", 0, 3, 4, 7))))
        return
    if args.policy_sweep:
        import subprocess
        for stat in (0, 1):
            for pol in range(8):
                env = dict(os.environ, FQ_POLICY_STAT=str(stat), FQ_POLICY_ONLINE=str(pol), FQ_POLICY_OFFLINE=str(pol))
                out = subprocess.run([sys.executable, __file__, "--only-headline", "--iters", str(args.iters)],
                                     env=env, capture_output=True, text=True).stdout
                keep = [l.strip() for l in out.splitlines() if "(" in l and "GB/s" in l and "torch" not in l]
                print("stat_nt=%d pol=%d (ntl=%d nts=%d rev=%d)" % (stat, pol, pol & 1, (pol >> 1) & 1, (pol >> 2) & 1))
                for l in keep:
                    print("    " + l)
        return
    dev = torch.device("cuda", 0)
    print(ops.device_info())
    shapes = SHAPES[:1] if args.only_headline else SHAPES
    for shape in shapes:
        torch.manual_seed(7)
        x = torch.relu(torch.randn(*shape, device=dev)) * 1.7
        y = torch.empty_like(x)
        numel = x.numel()
        thr = torch.tensor([4.0], device=dev)
        cur = torch.empty(1, device=dev)
        cases = {
            "absmax     (4B)": (lambda: ops.absmax_per_sample(x), 4),
            "offline    (8B)": (lambda: ops.fake_quant_offline(x, thr, 8, 0, out=y, want_stat=False), 8),
            "offl+stat  (8B)": (lambda: ops.fake_quant_offline(x, thr, 8, 0, out=y, cur_out=cur), 8),
            "online    (12B)": (lambda: ops.fake_quant_online(x, 8, 0, out=y, cur_out=cur), 12),
            "torch copy (8B)": (lambda: y.copy_(x), 8),
        }
        print("shape %s  %.1f MB" % (shape, numel * 4 / 1e6))
        for name, (fn, bpe) in cases.items():
            med, best = timeit(fn, args.iters)
            print("  %-16s median %8.3f ms  best %8.3f ms  -> %7.1f GB/s (best %7.1f)  %.1f%% of 8 TB/s"
                  % (name, med, best, bpe * numel / med / 1e6, bpe * numel / best / 1e6,
                     bpe * numel / med / 1e6 / 80.0))


if __name__ == "__main__":
    main()
