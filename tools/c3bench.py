#!/usr/bin/env python3
"""fq_conv3x3_i8 alone on ResNet-50's four stages (batch 128, online statistic, BatchNorm + ReLU + statistic on store): median of
40 launches each.  FQ_LIB_PATH selects a variant library; FQ_C3_DEEP / FQ_C3_NW / FQ_C3_PTW the dispatch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
out = []
for cin, hw in ((64, 56), (128, 28), (256, 14), (512, 7)):
    torch.manual_seed(7)
    x = torch.relu(torch.randn(128, cin, hw, hw, device=dev))
    w = torch.randn(cin, cin, 3, 3, device=dev) * 0.1
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
    stat = ops.absmax_per_sample(x)
    cur = torch.empty(1, device=dev)
    codes, scales, rowsum = ops.weight_codes_3x3(w, cin, 8)
    fn = lambda: ops.conv3x3_i8(x, codes, scales, rowsum, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc,
                                bn_shift=sh, act="relu")
    try:
        for _ in range(4):
            fn()
    except Exception as e:                      # (a forced variant the shape has no instantiation for)
        out.append("%d@%dx%d -" % (cin, hw, hw))
        continue
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    out.append("%d@%dx%d %.1f" % (cin, hw, hw, sorted(a.elapsed_time(b) for a, b in ev)[20] * 1e3))
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in ("FQ_C3_DEEP", "FQ_C3_NW", "FQ_C3_PTW") if k in os.environ)
print("%-30s %-24s | %s us" % (os.environ.get("FQ_LIB_PATH", "(default)")[-30:], tag, "  ".join(out)))
