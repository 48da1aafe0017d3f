"""`merge_bn(net, ...)` — reference: quantize/freeze/merge_bn.py:34-92: fold every BatchNorm that follows a Conv2D
into the conv's weight/bias once, then turn the BN into an identity.  Host-side, runs once per net; uses the generic
NDArray ops (this is model surgery, not the per-batch hot path)."""
import types

from ...mx import nd
from ...mx.gluon import nn

__all__ = ['merge_bn']


def _bypass_bn(net, exclude=[]):
    def _forward(self, F, x, *args, **kwargs):
        return x

    def _bypass(m):
        if isinstance(m, nn.BatchNorm) and not any(m is e for e in exclude):
            m.hybrid_forward = types.MethodType(_forward, m)
    net.apply(_bypass)


def _merge_bn(net, conv_name="conv", bn_name="batchnorm", exclude=[]):
    conv_lst = []

    def _collect_conv(m):
        if isinstance(m, nn.Conv2D):
            assert not hasattr(m, "gamma"), "Don't merge bn to a conv with fake bn! ({})".format(m.name)
            conv_lst.append(m)
    net.apply(_collect_conv)

    all_params = net.collect_params()
    for conv in conv_lst:
        bn = conv.name.replace(conv_name, bn_name)
        if bn + "_gamma" not in all_params or any(conv is e for e in exclude):
            continue
        print("Merge {} to {}".format(bn, conv.name))
        gamma = all_params[bn + "_gamma"].data()
        beta = all_params[bn + "_beta"].data()
        mean = all_params[bn + "_running_mean"].data()
        var = all_params[bn + "_running_var"].data()

        weight = conv.weight.data()
        w_shape = conv.weight.shape
        cout = w_shape[0]
        # NB the reference folds with 1e-10, not BatchNorm's own epsilon (merge_bn.py:64-65,73) — kept
        conv.weight.set_data((weight.reshape(cout, -1) * gamma.reshape(-1, 1)
                              / nd.sqrt(var + 1e-10).reshape(-1, 1)).reshape(w_shape))
        if conv.bias is None:
            conv._kwargs['no_bias'] = False
            conv.bias = conv.params.get('bias',
                                        shape=(cout,), init="zeros",
                                        allow_deferred_init=True)
            conv.bias.initialize(ctx=weight.context)
        bias = conv.bias.data()
        conv.bias.set_data(gamma * (bias - mean) / nd.sqrt(var + 1e-10) + beta)


def merge_bn(net, conv_name="conv", bn_name="batchnorm", exclude=[]):
    _merge_bn(net, conv_name, bn_name, exclude)
    _bypass_bn(net, exclude)
    return net
