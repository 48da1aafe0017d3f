#!/usr/bin/env python3
"""What the epilogue pieces of the fused convolution kernels cost: one layer of nn.Conv2D(quantized=True) (range mode: the
same pointwise / depthwise kernels as the simulated-quantisation path, but the threshold is ONE record load instead of the
batch-mean prologue) timed with and without the folded BatchNorm + ReLU and the per-sample statistic.

    python tools/qconv_ablate.py            (GPU box; median of 200 launches per variant, HIP events around each)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
LAYERS = [("pw 512->512 @14x14", 512, 512, 14, 1, 1, 0, 1), ("pw 1024->1024 @7x7", 1024, 1024, 7, 1, 1, 0, 1),
          ("pw 64->128 @56x56", 64, 128, 56, 1, 1, 0, 1), ("pw 32->64 @112x112", 32, 64, 112, 1, 1, 0, 1),
          ("dw 512 @14x14", 512, 512, 14, 3, 1, 1, 512), ("dw 128 @56x56", 128, 128, 56, 3, 1, 1, 128)]


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2] * 1e3


def main():
    torch.manual_seed(3)
    for name, cin, cout, hw, k, s, p, g in LAYERS:
        x = torch.relu(torch.randn(128, cin, hw, hw, device=dev)) * 2
        w = torch.randn(cout, cin // g, k, k, device=dev) * 0.1
        wbuf = ops.qconv_weights(w, (s, s), (p, p), g)
        ws = ops.qconv_workspace(cout, dev)
        stat = x.reshape(128, -1).amax(dim=1).contiguous()
        bsc, bsh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
        y = torch.empty(128, cout, hw, hw, device=dev)
        mode = "int8"                     # representable by construction: no fix-up launch in any variant
        base = dict(input_dtype=mode, in_stat=stat, out=y)
        res = []
        for label, kw in (("plain", {}), ("+relu", dict(act="relu")), ("+bn+relu", dict(act="relu", bn_scale=bsc, bn_shift=bsh)),
                          ("+bn+relu+stat", dict(act="relu", bn_scale=bsc, bn_shift=bsh, want_stat=True))):
            res.append((label, timed(lambda: ops.qconv2d(x, w, wbuf, None, (s, s), (p, p), g, ws, **base, **kw))))
        nbytes = 4.0 * (x.numel() + y.numel())
        print("%-22s " % name + "  ".join("%s %.1f us (%.2f)" % (l, t, nbytes / (t * 1e-6) / 8e12) for l, t in res))
    print("(each figure: finish kernel + convolution, median of 200; fraction of 8 TB/s on 4 B in + 4 B out)")


if __name__ == "__main__":
    main()
