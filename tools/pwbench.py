#!/usr/bin/env python3
"""Micro-benchmark of fq_pwconv_i8 on the 13 pointwise layers of mobilenet1.0 at batch 128 (vs torch fp32 conv)."""
import os
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

LAYERS = [(32, 64, 112), (64, 128, 56), (128, 128, 56), (128, 256, 28), (256, 256, 28), (256, 512, 14), (512, 512, 14),
          (512, 1024, 7), (1024, 1024, 7)]


def main():
    dev = torch.device("cuda", 0)
    n = 128
    tot, tot_ref, tot_b = 0.0, 0.0, 0.0
    mult = {(512, 512, 14): 5}
    for cin, cout, hw in LAYERS:
        torch.manual_seed(7)
        x = torch.relu(torch.randn(n, cin, hw, hw, device=dev))
        w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.randn(cout, device=dev)
        stat = ops.absmax_per_sample(x)
        cur = torch.empty(1, device=dev)
        codes, scales, rowsum = ops.weight_codes(w, cout, 8)
        nbytes = 4 * (x.numel() + n * cout * hw * hw)
        med, _ = timeit(lambda: ops.pwconv_i8(x, codes, scales, rowsum, in_stat=stat, width=8, flags=0, cur_out=cur,
                                              bn_scale=sc, bn_shift=sh, act="relu"), 20)
        ref, _ = timeit(lambda: TF.conv2d(x, w), 20)
        k = mult.get((cin, cout, hw), 1)
        tot += med * k
        tot_ref += ref * k
        tot_b += nbytes * k
        print("%4d->%4d %3dx%-3d %7.1f MB  int8 fused: %7.3f ms %7.1f GB/s    torch fp32 conv alone: %7.3f ms"
              % (cin, cout, hw, hw, nbytes / 1e6, med, nbytes / med / 1e6, ref))
    print("all 13 layers: int8 fused %.3f ms (%.1f GB/s)   torch conv alone %.3f ms" % (tot, tot_b / tot / 1e6, tot_ref))


if __name__ == "__main__":
    main()
