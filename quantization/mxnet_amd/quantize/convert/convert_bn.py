"""`bypass_bn(block)` — a BatchNorm becomes the identity (reference API: quantize/convert/convert_bn.py:32-36; used as
the `nn.BatchNorm` converter together with `fake_bn=True` / --merge-bn, where the convolution carries the fold)."""
from ...mx.gluon import nn
from ._blocks import passthrough, rebind_forward

__all__ = ['bypass_bn']


def bypass_bn(m):
    if not isinstance(m, nn.BatchNorm):
        raise AssertionError("bypass_bn expects a BatchNorm, got %s" % type(m).__name__)
    rebind_forward(m, passthrough, keep_origin=False)
