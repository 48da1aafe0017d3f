#!/usr/bin/env python3
"""Experiment: evaluation steps of INDEPENDENT batches on S HIP streams (one replica of the net per stream, same weights):
do the ramps and tails of one step's kernels fill with the other step's work?  python tools/streams_probe.py [S ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from quantization.mxnet_amd import mx, ops  # noqa: E402
from quantization.mxnet_amd.quantize import fuse  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = mx.gpu(0)
    counts = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 3, 1, 2]
    smax = max(counts)
    if "--one-net" in sys.argv:                          # forwards of ONE net in flight (per-stream arenas and slots) ...
        nets = [bench.build_net("mobilenet1.0", 1000, ctx)] * smax
    else:                                               # ... or a replica per stream
        nets = [bench.build_net("mobilenet1.0", 1000, ctx) for _ in range(smax)]
    torch.manual_seed(7)
    batches = [mx.nd.NDArray(torch.randn(128, 3, 224, 224, device=dev)) for _ in range(4)]
    labels = [torch.randint(0, 1000, (128,), device=dev) for _ in range(4)]
    counters = torch.zeros(2002, device=dev)
    heads = [fuse.eval_head(n, counters) for n in nets]
    if "--one-net" in sys.argv:
        heads = [heads[-1]] * smax
    streams = [torch.cuda.Stream(dev) for _ in range(smax)]

    def step(i, s):
        with torch.cuda.stream(streams[s]), ops.batches_in_flight():
            if heads[s] is not None:
                heads[s].labels = labels[i % 4]
            out = nets[s](batches[i % 4])._t
            if heads[s] is None or not heads[s].take():
                ops.eval_counters(out, labels[i % 4], counters)

    for s in range(smax):
        for i in range(6):
            step(i, s)
    torch.cuda.synchronize()
    for S in counts:
        steps = 600
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, i % S)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("streams %d: %.4f ms/step  %.0f images/s" % (S, dt / steps * 1e3, steps * 128 / dt))


if __name__ == "__main__":
    main()
