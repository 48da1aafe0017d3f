#!/usr/bin/env python3
"""fq_pwconv_i8_c16_dual alone on the closing 1x1 convolutions of ResNet-50's four stages (batch 128, codes in, residual operand,
fp32 out + the code copy), against the same call with ONE output: median of 40 launches.  FQ_LIB_PATH selects a variant library."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
out = []
for cin, cout, hw in ((64, 256, 56), (128, 512, 28), (256, 1024, 14), (512, 2048, 7)):
    torch.manual_seed(3)
    thr = torch.tensor([2.3], device=dev)
    xc = ops.Codes16(torch.randint(-128, 127, (128, (cin + 15) // 16, hw * hw, 16), dtype=torch.int8, device=dev),
                     (128, cin, hw, hw), thr, 8, 0)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
    codes, scales, rowsum = ops.weight_codes(w, 1, 8)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    res = torch.randn(128, cout, hw, hw, device=dev)
    stat_in = torch.rand(128, device=dev) + 2
    kw = dict(in_thr=thr, width=8, flags=0, bn_scale=sc, bn_shift=sh, act="relu", in_stat=stat_in, residual=res)
    side = dict(thr=torch.tensor([3.1], device=dev), width=8, flags=0)

    def timeit(fn):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)[20] * 1e3
    one = timeit(lambda: ops.pwconv_i8(xc, codes, scales, rowsum, **kw))
    two = timeit(lambda: ops.pwconv_i8(xc, codes, scales, rowsum, side_codes=side, **kw))
    out.append("%d->%d@%d %.1f / %.1f" % (cin, cout, hw, one, two))
print("%-32s | one output / two (us): %s" % (os.environ.get("FQ_LIB_PATH", "(default)")[-32:], "   ".join(out)))
