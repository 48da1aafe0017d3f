#!/usr/bin/env python3
"""fp32 1x1 convolutions of ResNet-50 (batch 128) through the tensor library: `conv2d` (MIOpen picks an NHWC implicit GEMM with
transposes around it for most of them) against a batched GEMM on the NCHW tensor as it lies (`matmul(W, x.view(n, Cin, HW))`).
The un-quantised forward of the KL collection runs 36 such layers per batch."""
import torch
import torch.nn.functional as TF

dev = torch.device("cuda", 0)
SHAPES = [(64, 64, 56, 1), (64, 256, 56, 1), (256, 64, 56, 1), (256, 128, 56, 2), (256, 512, 56, 2), (128, 512, 28, 1), (512, 128, 28, 1),
          (512, 256, 28, 2), (512, 1024, 28, 2), (256, 1024, 14, 1), (1024, 256, 14, 1), (1024, 512, 14, 2), (1024, 2048, 14, 2),
          (512, 2048, 7, 1), (2048, 512, 7, 1)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2] * 1e3


tot = [0.0, 0.0]
for cin, cout, hw, s in SHAPES:
    x = torch.randn(128, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    w2 = w.view(cout, cin)

    def conv():
        return TF.conv2d(x, w, None, stride=s)

    def gemm():
        xs = x if s == 1 else x[:, :, ::s, ::s].contiguous()
        n, c, h, ww = xs.shape
        return torch.matmul(w2, xs.view(n, c, h * ww)).view(n, cout, h, ww)
    a, b = conv(), gemm()
    err = float((a - b).abs().max() / a.abs().max())
    ta, tb = timed(conv), timed(gemm)
    tot[0] += ta
    tot[1] += tb
    print("%4d -> %4d @%2dx%-2d stride %d   conv2d %7.1f us   batched GEMM %7.1f us   max |diff| / max |y| %.1e" % (cin, cout, hw, hw, s, ta, tb, err))
print("sum %.1f us against %.1f us" % (tot[0], tot[1]))
