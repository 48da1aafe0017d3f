"""`qparams_init(net, conv_name="conv", bn_name="batchnorm")` — behaviour of the reference's
quantize/initialize/initialize.py:31-75, written on this project's block machinery (convert/_blocks.py).

Every calibrated range starts at 0 (so the naive-EMA estimate carries the reference's (1 - 0.9^k) bias).  A convolution
converted with `fake_bn=True` adopts the vectors of the BatchNorm that follows it — found by the reference's naming rule
`conv.name.replace(conv_name, bn_name)` — and gains a zero bias when it had none.
"""
from ...mx.gluon import nn
from ...mx.initializer import Constant
from ..convert._blocks import BatchNormTerms, ensure_bias

__all__ = ["qparams_init"]


def _home_ctx(block):
    """context of the block's weight when it already has one (ranges are created next to it)"""
    weight = getattr(block, "weight", None)
    if weight is None:
        return None
    try:
        return weight.list_ctx()[0]
    except Exception:
        return None


def _adopt_batchnorm(conv, every_param, conv_name, bn_name):
    terms = BatchNormTerms.of_sibling(conv, every_param, conv_name, bn_name)
    if terms is None:
        raise KeyError("fake_bn: no BatchNorm named %s* for convolution %s"
                       % (conv.name.replace(conv_name, bn_name), conv.name))
    ctx = _home_ctx(conv)
    for field, value in zip(BatchNormTerms.FIELDS, (terms.gamma, terms.beta, terms.mean, terms.var)):
        getattr(conv, field).initialize(Constant(value), ctx=ctx)
    ensure_bias(conv, ctx)


def qparams_init(net, conv_name="conv", bn_name="batchnorm"):
    every_param = net.collect_params()
    for block in net.collect_quantized_blocks():
        kind = type(block)
        if kind is nn.Activation:
            if block.quantize_args.quantize_act:
                block.act_max.initialize(Constant(0))
            continue
        if isinstance(block, nn.Conv2D) and hasattr(block, "gamma"):
            _adopt_batchnorm(block, every_param, conv_name, bn_name)
        if kind in (nn.Conv2D, nn.Dense) and block.quantize_args.quantize_input:
            block.input_max.initialize(Constant(0), ctx=_home_ctx(block))
    return net
