#!/usr/bin/env python3
"""The first convolutions alone (batch 128, 224x224): 3x3 -> 32 (MobileNets) and 7x7 -> 64 (ResNets), median of 40 launches.
FQ_LIB_PATH selects a variant library (csrc/build.py --only fq_stem -DFQ_STEM_CH=..); FQ_STEM_WG_PER_CU the grid.  (r4: two to
four workgroups per CU, 8 or 16 loads in flight, 128 / 168 / 256 registers: 323-375 us for the 7x7 form, none better than the
default's 324; the 3x3 form 83-87 us.  `-DFQ_STEM_NOSTORE=1`: the statistic without the stores - 53.3 us for the 3x3 form, 297 us
for the 7x7 form: what a recomputation of these layers would cost)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
out = []
for ks, cout in ((3, 32), (7, 64)):
    x = torch.randn(128, 3, 224, 224, device=dev)
    w = torch.randn(cout, 3, ks, ks, device=dev) * 0.2
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    fn = lambda: ops.stem_conv_s2(x, w, None, bn_scale=sc, bn_shift=sh, act="relu")
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    out.append("%dx%d -> %d: %.1f us" % (ks, ks, cout, sorted(a.elapsed_time(b) for a, b in ev)[20] * 1e3))
# round 5: the ResNet head in one launch (fq_stem_conv7x7s2_pool) against its two launches (convolution, then pooling + statistic)
x = torch.randn(128, 3, 224, 224, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) * 0.2
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
one, zero = torch.ones(64, device=dev), torch.zeros(64, device=dev)


def two():
    y, _ = ops.stem_conv_s2(x, w, None, bn_scale=sc, bn_shift=sh, act="relu")
    return ops.bn_act_maxpool_stat(y, one, zero, "none", want_stat=True)


for name, fn in (("7x7 -> 64 + max-pool, two launches", two),
                 ("7x7 -> 64 + max-pool, one launch", lambda: ops.stem_conv_s2(x, w, None, bn_scale=sc, bn_shift=sh, act="relu", pool=True))):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    out.append("%s: %.1f us" % (name, sorted(a.elapsed_time(b) for a, b in ev)[20] * 1e3))
print("%-30s wg/cu %s | %s" % (os.environ.get("FQ_LIB_PATH", "(default)")[-30:], os.environ.get("FQ_STEM_WG_PER_CU", "-"),
                              "  ".join(out)))
