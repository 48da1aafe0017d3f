#!/usr/bin/env python3
"""Micro-benchmark of fq_dwconv3x3 / fq_bn_act_stat on the 13 depthwise layers of mobilenet1.0 at batch 128."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

LAYERS = [(32, 112, 1), (64, 112, 2), (128, 56, 1), (128, 56, 2), (256, 28, 1), (256, 28, 2), (512, 14, 1),
          (512, 14, 2), (1024, 7, 1)]


def main():
    dev = torch.device("cuda", 0)
    n = 128
    tot_t, tot_b = 0.0, 0.0
    mult = {(512, 14, 1): 5}
    for c, hw, s in LAYERS:
        torch.manual_seed(7)
        x = torch.relu(torch.randn(n, c, hw, hw, device=dev))
        w = torch.randn(c, 1, 3, 3, device=dev) * 0.3
        sc = torch.rand(c, device=dev) + 0.5
        sh = torch.randn(c, device=dev)
        stat = ops.absmax_per_sample(x)
        cur = torch.empty(1, device=dev)
        ho = (hw - 1) // s + 1
        nbytes = 4 * (x.numel() + n * c * ho * ho)
        med, best = timeit(lambda: ops.dwconv3x3(x, w, None, stride=s, in_stat=stat, width=8, flags=0, cur_out=cur,
                                                 bn_scale=sc, bn_shift=sh, act="relu"), 20)
        med2, _ = timeit(lambda: ops.dwconv3x3(x, w, None, stride=s, bn_scale=sc, bn_shift=sh, act="relu"), 20)
        k = mult.get((c, hw, s), 1)
        tot_t += med * k
        tot_b += nbytes * k
        print("C=%4d %3dx%-3d s%d  %7.1f MB  quant+bn+relu+stat: %7.3f ms %7.1f GB/s   no-quant: %7.3f ms %7.1f GB/s"
              % (c, hw, hw, s, nbytes / 1e6, med, nbytes / med / 1e6, med2, nbytes / med2 / 1e6))
    print("all 13 layers: %.3f ms, %.1f GB/s" % (tot_t, tot_b / tot_t / 1e6))


if __name__ == "__main__":
    main()
