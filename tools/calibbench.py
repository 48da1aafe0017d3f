#!/usr/bin/env python3
"""Calibration-kernel micro-benchmark (not the driver's bench.py): `fq_global_max`, `fq_histogram_accumulate` and
`fq_kl_search` timed with HIP events on torch's current stream.

    python tools/calibbench.py [--iters 20] [--json profiles/r2_calibbench.json]

Histogram inputs follow SURVEY 8(d): post-ReLU-like max(N(0,1),0)*sigma (about half exact zeros), seed 7, at the
shapes whose per-batch activations dominate MobileNet / ResNet-50 calibration.  KL inputs are the histograms of such
tensors, L = 27 (MobileNet) and 53 (ResNet-50) layers in ONE launch, levels 256 (unsigned 8 bit) and 128 (signed)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

SHAPES = [(128, 64, 112, 112), (128, 256, 56, 56), (128, 128, 56, 56), (128, 256, 28, 28), (128, 512, 14, 14),
          (128, 1024, 7, 7), (128, 2048)]


def timeit(fn, iters, warmup=3):
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
    ap.add_argument("--json", default=None)
    ap.add_argument("--skip-kl", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    info = ops.device_info()
    print(info)
    rec = {"device": info, "histogram": [], "global_max": [], "kl_search": [],
           "env": {k: v for k, v in os.environ.items() if k.startswith("FQ_")}}
    hists = []
    for shape in SHAPES:
        torch.manual_seed(7)
        x = torch.relu(torch.randn(*shape, device=dev)) * 1.7
        numel = x.numel()
        zeros = float((x == 0).float().mean().item())
        mx = ops.global_max(x)
        hist = torch.zeros(2048, dtype=torch.int64, device=dev)
        ops.histogram_accumulate(x, mx, hist)
        torch.cuda.synchronize()
        assert int(hist.sum().item()) == int((x != 0).sum().item()), "histogram mass"
        hists.append(ops.hist_to_float(hist))
        for name, fn, key in (("global_max", lambda: ops.global_max(x), "global_max"),
                              ("histogram ", lambda: ops.histogram_accumulate(x, mx, hist), "histogram")):
            med, best = timeit(fn, args.iters)
            gbs = 4 * numel / med / 1e6
            print("%-26s %s  %7.1f MB  zeros %.2f  median %8.3f ms  best %8.3f ms -> %7.1f GB/s  (%.1f%% of 8 TB/s)"
                  % (str(shape), name, numel * 4 / 1e6, zeros, med, best, gbs, gbs / 80.0))
            rec[key].append({"shape": list(shape), "mbytes": numel * 4 / 1e6, "zeros": zeros, "median_ms": med,
                             "best_ms": best, "gbps": gbs, "frac_of_8TBps": gbs / 8000.0})
    if not args.skip_kl:
        for L in (27, 53):
            h = torch.stack([hists[i % len(hists)] * (1.0 + 0.01 * i) for i in range(L)]).contiguous()
            for levels in (256, 128):
                med, best = timeit(lambda: ops.kl_search(h, levels, levels), max(3, args.iters // 4), warmup=1)
                print("kl_search  L=%d levels=%d bins=2048: median %8.3f ms  best %8.3f ms  (%.3f ms / layer)"
                      % (L, levels, med, best, med / L))
                rec["kl_search"].append({"L": L, "levels": levels, "bins": 2048, "median_ms": med, "best_ms": best})
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
