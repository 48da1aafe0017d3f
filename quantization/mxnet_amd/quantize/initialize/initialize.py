"""`qparams_init` — reference: quantize/initialize/initialize.py:31-75.

`input_max` / `act_max` start at 0 (so the naive-EMA estimate carries the (1 - 0.9^k) bias of the reference);
with fake-BN the conv adopts its sibling BatchNorm's gamma/beta/running stats, found by name
(`name.replace(conv_name, bn_name)`), and gains a bias if it had none."""
from ...mx.gluon import nn
from ...mx.initializer import Constant

__all__ = ["qparams_init"]


def qparams_init(net, conv_name="conv", bn_name="batchnorm"):
    blocks = net.collect_quantized_blocks()
    params = net.collect_params()

    for m in blocks:
        # If fake bn, recalculate weight and initialize some related params (:46-70)
        if isinstance(m, nn.Conv2D) and hasattr(m, "gamma"):
            name = m.name

            # Get params of batchnorm
            gamma = params[name.replace(conv_name, bn_name) + "_gamma"].data()
            beta = params[name.replace(conv_name, bn_name) + "_beta"].data()
            mean = params[name.replace(conv_name, bn_name) + "_running_mean"].data()
            var = params[name.replace(conv_name, bn_name) + "_running_var"].data()
            ctx = m.weight.list_ctx()[0]

            # Store params of bn at conv
            m.gamma.initialize(Constant(gamma), ctx=ctx)
            m.beta.initialize(Constant(beta), ctx=ctx)
            m.running_mean.initialize(Constant(mean), ctx=ctx)
            m.running_var.initialize(Constant(var), ctx=ctx)

            # Enable bias if need
            cout = m.weight.shape[0]
            if m.bias is None:
                m._kwargs['no_bias'] = False
                m.bias = m.params.get('bias',
                                      shape=(cout,), init="zeros",
                                      allow_deferred_init=True)
                m.bias.initialize(ctx=ctx)

        if type(m) in (nn.Conv2D, nn.Dense) and m.quantize_args.quantize_input:
            m.input_max.initialize(Constant(0), ctx=_ctx_of(m))
        if type(m) == nn.Activation and m.quantize_args.quantize_act:
            m.act_max.initialize(Constant(0))
    return net


def _ctx_of(m):
    w = getattr(m, "weight", None)
    try:
        return w.list_ctx()[0] if w is not None else None
    except Exception:
        return None
