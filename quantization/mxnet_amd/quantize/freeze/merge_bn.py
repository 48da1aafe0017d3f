"""`merge_bn(net, conv_name="conv", bn_name="batchnorm", exclude=[])` — behaviour of the reference's
quantize/freeze/merge_bn.py:34-92: fold every BatchNorm that follows a Conv2D (same naming rule as `qparams_init`) into
the convolution's weight and bias once, then turn the BatchNorms into identities.  Model surgery on the host side, runs
once per net; the arithmetic lives in convert/_blocks.py (`BatchNormTerms`, folding with sqrt(var + 1e-10) like the
reference, not with the BatchNorm's own epsilon)."""
from ...mx import nd
from ...mx.gluon import nn
from ..convert._blocks import BatchNormTerms, ensure_bias, passthrough, rebind_forward

__all__ = ['merge_bn']


def _is_excluded(block, exclude):
    return any(block is e for e in exclude)


def merge_bn(net, conv_name="conv", bn_name="batchnorm", exclude=[]):
    convs = []

    def gather(block):
        if isinstance(block, nn.Conv2D):
            if hasattr(block, "gamma"):
                raise AssertionError("Don't merge bn to a conv with fake bn! ({})".format(block.name))
            convs.append(block)
    net.apply(gather)

    every_param = net.collect_params()
    for conv in convs:
        terms = None if _is_excluded(conv, exclude) else BatchNormTerms.of_sibling(conv, every_param, conv_name, bn_name)
        if terms is None:
            continue
        print("Merge {} to {}".format(conv.name.replace(conv_name, bn_name), conv.name))
        weight = conv.weight.data()
        conv.weight.set_data(terms.fold_weight(nd, weight))
        bias = ensure_bias(conv, weight.context)
        bias.set_data(terms.fold_bias(nd, bias.data()))

    def silence(block):
        if isinstance(block, nn.BatchNorm) and not _is_excluded(block, exclude):
            rebind_forward(block, passthrough, keep_origin=False)
    net.apply(silence)
    return net
