#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950):
a plain device copy of known size (calibrates the counters' units/corrections for THIS access pattern), then the
fake-quant entry points on the headline tensor (128,64,112,112).  See profiles/README.md for the commands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(7)
x = torch.relu(torch.randn(128, 64, 112, 112, device=dev)) * 1.7
y = torch.empty_like(x)
thr = torch.tensor([4.0], device=dev)
cur = torch.empty(1, device=dev)
hist = torch.zeros(2048, dtype=torch.int64, device=dev)
mx = ops.global_max(x)
for _ in range(3):
    y.copy_(x)                                                      # calibration: 411 MB read + 411 MB written
    ops.fake_quant_offline(x, thr, 8, 0, out=y, want_stat=False)    # apply only
    ops.fake_quant_online(x, 8, 0, out=y, cur_out=cur)              # statistic pass + apply pass
    ops.histogram_accumulate(x, mx, hist)
torch.cuda.synchronize()
print("numel", x.numel(), "bytes", x.numel() * 4)
