#!/usr/bin/env python3
"""Which Python call sites issue device-to-device copies / clones during one fused benchmark forward?"""
import collections
import os
import sys
import traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from quantization.mxnet_amd import mx  # noqa: E402
ctx = mx.gpu(0)
net = bench.build_net("mobilenet1.0", 1000, ctx, fuse=True)
X = mx.nd.NDArray(torch.randn(128, 3, 224, 224, device=ctx.torch_device))
for _ in range(3):
    net(X)
sites = collections.Counter()


def wrap(name, fn):
    def w(self, *a, **k):
        if self.is_cuda:
            st = [f for f in traceback.extract_stack()[:-1] if "/root/repo/" in f.filename or "repo/" in f.filename]
            key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-3:])
            sites[(name, key)] += 1
        return fn(self, *a, **k)
    return w


for name in ("copy_", "clone", "contiguous", "to", "fill_", "zero_"):
    setattr(torch.Tensor, name, wrap(name, getattr(torch.Tensor, name)))
net(X)
torch.cuda.synchronize()
for (name, key), c in sites.most_common(25):
    print("%3d  %-10s %s" % (c, name, key))
