#!/usr/bin/env python3
"""fq_dwconv3x3_c16 alone on MobileNetV2's depthwise layers (batch 128): median of 60 launches each, HIP events.
FQ_LIB_PATH selects a variant library (csrc/build.py --only fq_dwconv16 -DFQ_DW16_V=<bits>); DW16_BATCH another batch size
(how much of a layer's time is its last, partly filled round of workgroups), DW16_LAYERS="2,3" a subset of the layers."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
LAYERS = [(32, 112, 1), (96, 112, 2), (144, 56, 1), (144, 56, 2), (192, 28, 1), (192, 28, 2), (384, 14, 1), (576, 14, 1),
          (576, 14, 2), (960, 7, 1)]
tot = 0.0
out = []
NB = int(os.environ.get("DW16_BATCH", "128"))
if os.environ.get("DW16_LAYERS"):
    LAYERS = [LAYERS[int(i)] for i in os.environ["DW16_LAYERS"].split(",")]
H_OVERRIDE = int(os.environ.get("DW16_H", "0"))       # planes of this many rows (same width): what a launch costs besides its rows
for c, hw, s in LAYERS:
    cb = (c + 15) // 16
    hh = H_OVERRIDE or hw
    x = torch.randint(-128, 127, (NB, cb, hh * hw, 16), dtype=torch.int8, device=dev)
    thr = torch.tensor([3.0], device=dev)
    xc = ops.Codes16(x, (NB, c, hh, hw), thr, 8, 0)
    w = torch.randn(c, 1, 3, 3, device=dev) * 0.3
    bsc, bsh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    oc = dict(thr=torch.tensor([2.5], device=dev), width=8, flags=0)
    fn = lambda: ops.dwconv3x3_c16(xc, w, None, stride=s, in_thr=thr, width=8, flags=0, bn_scale=bsc, bn_shift=bsh,
                                   act="relu6", out_codes=oc)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)[30] * 1e3
    tot += t
    out.append("%dx%d/s%d %.1f" % (c, hw, s, t))
print("%-28s batch %d sum %.1f us | " % (os.environ.get("FQ_LIB_PATH", "(default)")[-28:], NB, tot) + "  ".join(out))
