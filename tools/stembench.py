#!/usr/bin/env python3
"""Timing of fq_stem_conv3x3s2 at the benchmark shape (128, 3, 224, 224) -> (128, 32, 112, 112)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(1)
x = torch.randn(128, 3, 224, 224, device=dev)
w = torch.randn(32, 3, 3, 3, device=dev) * 0.2
wt = w.permute(1, 2, 3, 0).contiguous()
sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev)
med, _ = timeit(lambda: ops.stem_conv3x3s2(x, w, bn_scale=sc, bn_shift=sh, act="relu", w_tap_major=wt), 30)
nbytes = 4 * (x.numel() + 128 * 32 * 112 * 112)
print("stem 3->32 s2 @224: %.1f us  %.0f GB/s" % (med * 1e3, nbytes / med / 1e6))
