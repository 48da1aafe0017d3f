#!/usr/bin/env python3
"""Which form of fq_pwconv_i8 is fastest on each pointwise layer of mobilenet1.0 (batch 128, online statistic, BN + ReLU)?
auto = the library's own choice; a form that does not take the shape prints '-'.  Median of 30 launches by events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

LAYERS = [(32, 64, 112), (64, 128, 56), (128, 128, 56), (128, 256, 28), (256, 256, 28), (256, 512, 14), (512, 512, 14),
          (512, 1024, 7), (1024, 1024, 7)]
if "--resnet" in sys.argv:          # resnet50_v1's 1x1 layers: (Cin, Cout, plane, residual operand)
    LAYERS = [(64, 64, 56, 0), (64, 256, 56, 1), (256, 64, 56, 0), (256, 128, 56, 0), (128, 512, 28, 1), (512, 128, 28, 0),
              (512, 256, 28, 0), (256, 1024, 14, 1), (1024, 256, 14, 0), (1024, 512, 14, 0), (512, 2048, 7, 1), (2048, 512, 7, 0)]
dev = torch.device("cuda", 0)
n = 128
if "--mobilenetv2" in sys.argv:     # mobilenetv2_1.0's 1x1 layers (expansion, projection with / without the unit's shortcut)
    LAYERS = [(32, 32, 112, 0), (32, 16, 112, 0), (16, 96, 112, 0), (96, 24, 56, 0), (24, 144, 56, 0), (144, 24, 56, 1),
              (144, 32, 28, 0), (32, 192, 28, 0), (192, 32, 28, 1), (192, 64, 14, 0), (64, 384, 14, 0), (384, 64, 14, 1),
              (384, 96, 14, 0), (96, 576, 14, 0), (576, 96, 14, 1), (576, 160, 7, 0), (160, 960, 7, 0), (960, 160, 7, 1),
              (960, 320, 7, 0), (320, 1280, 7, 0)]
for layer in LAYERS:
    cin, cout, hw = layer[:3]
    res = torch.randn(n, cout, hw, hw, device=dev) if len(layer) > 3 and layer[3] else None
    torch.manual_seed(7)
    x = torch.relu(torch.randn(n, cin, hw, hw, device=dev))
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    stat = ops.absmax_per_sample(x)
    cur = torch.empty(1, device=dev)
    codes, scales, rowsum = ops.weight_codes(w, cout, 8)
    out = []
    for form in (None, "stream", "sample", "split", "pipe"):
        kw = dict(in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc, bn_shift=sh, act="relu")
        if res is not None:
            kw["residual"] = res
        if form:
            kw["form"] = form
        try:
            fn = lambda: ops.pwconv_i8(x, codes, scales, rowsum, **kw)
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
            for a, b in ev:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
            out.append("%s %6.1f" % (form or "auto", sorted(a.elapsed_time(b) for a, b in ev)[15] * 1e3))
        except Exception:
            out.append("%s      -" % (form or "auto"))
    print("%4d->%4d @%3d%s: " % (cin, cout, hw, " +res" if res is not None else "     ") + "   ".join(out) + "  us")
