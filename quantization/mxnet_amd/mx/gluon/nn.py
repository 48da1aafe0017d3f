"""`gluon.nn` layers used by the reference's model families (gluoncv mobilenet / mobilenetv2 / resnet_v1 /
cifar_resnet_v1) with the attributes the converters read: `Conv2D._kwargs['kernel'|'num_filter'|'num_group'|'no_bias']`
(quantize/convert/convert_conv2d.py:71,74,85; initialize.py:66), `Dense._units` (convert_dense.py:53),
`Activation._act_type` (convert_act.py:36,40).  Convolution / FC / BN / pooling themselves run through torch
(MIOpen / rocBLAS): they are not the path this project replaces.
"""
from .block import Block, HybridBlock
from .. import initializer as _init

__all__ = ["Block", "HybridBlock", "Sequential", "HybridSequential", "Conv2D", "Dense", "Activation", "BatchNorm",
           "MaxPool2D", "AvgPool2D", "GlobalAvgPool2D", "GlobalMaxPool2D", "Flatten", "Dropout", "HybridLambda"]


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Sequential(Block):
    def __init__(self, prefix=None, params=None):
        super(Sequential, self).__init__(prefix=prefix, params=params)

    def add(self, *blocks):
        for b in blocks:
            self.register_child(b)

    def forward(self, x):
        for b in self._children.values():
            x = b(x)
        return x

    def __getitem__(self, key):
        layers = list(self._children.values())[key]
        if isinstance(layers, list):
            net = type(self)(prefix=self._prefix)
            net.add(*layers)
            return net
        return layers

    def __len__(self):
        return len(self._children)

    def __iter__(self):
        return iter(self._children.values())


class HybridSequential(HybridBlock):
    def __init__(self, prefix=None, params=None):
        super(HybridSequential, self).__init__(prefix=prefix, params=params)

    def add(self, *blocks):
        for b in blocks:
            self.register_child(b)

    def forward(self, x):
        for b in self._children.values():
            x = b(x)
        return x

    def hybrid_forward(self, F, x):
        return self.forward(x)

    __getitem__ = Sequential.__getitem__
    __len__ = Sequential.__len__
    __iter__ = Sequential.__iter__


class Activation(HybridBlock):
    def __init__(self, activation, **kwargs):
        self._act_type = activation
        super(Activation, self).__init__(**kwargs)

    def _alias(self):
        return self._act_type

    def hybrid_forward(self, F, x):
        return F.Activation(x, act_type=self._act_type, name="fwd")

    def __repr__(self):
        return "Activation(%s)" % self._act_type


class Conv2D(HybridBlock):
    def __init__(self, channels, kernel_size, strides=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1,
                 layout="NCHW", activation=None, use_bias=True, weight_initializer=None,
                 bias_initializer="zeros", in_channels=0, **kwargs):
        super(Conv2D, self).__init__(**kwargs)
        assert layout == "NCHW", "Only supports 'NCHW' layout for now"
        with self.name_scope():
            self._channels = channels
            self._in_channels = in_channels
            kernel_size = _pair(kernel_size)
            self._kwargs = {
                "kernel": kernel_size, "stride": _pair(strides), "dilate": _pair(dilation),
                "pad": _pair(padding), "num_filter": channels, "num_group": groups,
                "no_bias": not use_bias, "layout": layout}
            wshape = (channels, in_channels // groups if in_channels else 0) + kernel_size
            self.weight = self.params.get("weight", shape=wshape, init=weight_initializer,
                                          allow_deferred_init=True)
            if use_bias:
                self.bias = self.params.get("bias", shape=(channels,), init=bias_initializer,
                                            allow_deferred_init=True)
            else:
                self.bias = None
            if activation is not None:
                self.act = Activation(activation, prefix=activation + "_")
            else:
                self.act = None

    def _alias(self):
        return "conv"

    def _infer_param_shapes(self, x, *args):
        cin = x.shape[1]
        g = self._kwargs["num_group"]
        self._in_channels = cin
        self.weight._finish_deferred_init((self._channels, cin // g) + tuple(self._kwargs["kernel"]))
        for p in self._reg_params.values():
            if p._data is None and p._deferred is not None and p._shape_known():
                p._finish_deferred_init(p.shape)

    def hybrid_forward(self, F, x, weight, bias=None):
        act = F.Convolution(x, weight, bias, name="fwd", **self._kwargs)
        if self.act is not None:
            act = self.act(act)
        return act

    def __repr__(self):
        k = self._kwargs
        shape = self.weight.shape
        return "Conv2D(%s -> %s, kernel_size=%s, stride=%s, padding=%s%s%s)" % (
            (shape[1] * k["num_group"]) if shape and shape[1] else None, shape[0], k["kernel"], k["stride"], k["pad"],
            ", groups=%d" % k["num_group"] if k["num_group"] != 1 else "", ", bias=False" if self.bias is None else "")


class Dense(HybridBlock):
    def __init__(self, units, activation=None, use_bias=True, flatten=True, dtype="float32",
                 weight_initializer=None, bias_initializer="zeros", in_units=0, **kwargs):
        super(Dense, self).__init__(**kwargs)
        self._flatten = flatten
        with self.name_scope():
            self._units = units
            self._in_units = in_units
            self.weight = self.params.get("weight", shape=(units, in_units), init=weight_initializer,
                                          allow_deferred_init=True)
            if use_bias:
                self.bias = self.params.get("bias", shape=(units,), init=bias_initializer,
                                            allow_deferred_init=True)
            else:
                self.bias = None
            if activation is not None:
                self.act = Activation(activation, prefix=activation + "_")
            else:
                self.act = None

    def _infer_param_shapes(self, x, *args):
        n = 1
        for s in (x.shape[1:] if self._flatten else x.shape[-1:]):
            n *= s
        self._in_units = n
        self.weight._finish_deferred_init((self._units, n))
        for p in self._reg_params.values():
            if p._data is None and p._deferred is not None and p._shape_known():
                p._finish_deferred_init(p.shape)

    def hybrid_forward(self, F, x, weight, bias=None):
        act = F.FullyConnected(x, weight, bias, no_bias=bias is None, num_hidden=self._units,
                               flatten=self._flatten, name="fwd")
        if self.act is not None:
            act = self.act(act)
        return act

    def __repr__(self):
        shape = self.weight.shape
        return "Dense(%s -> %s, %s)" % (shape[1] if shape[1] else None, shape[0],
                                        self.act if self.act else "linear")


class BatchNorm(HybridBlock):
    def __init__(self, axis=1, momentum=0.9, epsilon=1e-5, center=True, scale=True, use_global_stats=False,
                 beta_initializer="zeros", gamma_initializer="ones", running_mean_initializer="zeros",
                 running_variance_initializer="ones", in_channels=0, **kwargs):
        super(BatchNorm, self).__init__(**kwargs)
        self._kwargs = {"axis": axis, "eps": epsilon, "momentum": momentum, "fix_gamma": not scale,
                        "use_global_stats": use_global_stats}
        self._in_channels = in_channels
        self.gamma = self.params.get("gamma", grad_req="write" if scale else "null", shape=(in_channels,),
                                     init=gamma_initializer, allow_deferred_init=True, differentiable=scale)
        self.beta = self.params.get("beta", grad_req="write" if center else "null", shape=(in_channels,),
                                    init=beta_initializer, allow_deferred_init=True, differentiable=center)
        self.running_mean = self.params.get("running_mean", grad_req="null", shape=(in_channels,),
                                            init=running_mean_initializer, allow_deferred_init=True,
                                            differentiable=False)
        self.running_var = self.params.get("running_var", grad_req="null", shape=(in_channels,),
                                           init=running_variance_initializer, allow_deferred_init=True,
                                           differentiable=False)

    def _infer_param_shapes(self, x, *args):
        c = x.shape[self._kwargs["axis"]]
        self._in_channels = c
        for p in (self.gamma, self.beta, self.running_mean, self.running_var):
            if p._data is None:
                p._finish_deferred_init((c,))

    def hybrid_forward(self, F, x, gamma, beta, running_mean, running_var):
        return F.BatchNorm(x, gamma, beta, running_mean, running_var, name="fwd", **self._kwargs)

    def __repr__(self):
        return "BatchNorm(%s, in_channels=%s)" % (
            ", ".join("%s=%s" % kv for kv in self._kwargs.items()), self.gamma.shape[0] or None)


class _Pooling(HybridBlock):
    def __init__(self, pool_size, strides, padding, ceil_mode, global_pool, pool_type, count_include_pad=None,
                 **kwargs):
        super(_Pooling, self).__init__(**kwargs)
        if strides is None:
            strides = pool_size
        self._kwargs = {"kernel": _pair(pool_size), "stride": _pair(strides), "pad": _pair(padding),
                        "global_pool": global_pool, "pool_type": pool_type,
                        "pooling_convention": "full" if ceil_mode else "valid"}
        if count_include_pad is not None:
            self._kwargs["count_include_pad"] = count_include_pad

    def _alias(self):
        return "pool"

    def hybrid_forward(self, F, x):
        return F.Pooling(x, name="fwd", **self._kwargs)

    def __repr__(self):
        return "%s(%s)" % (self.__class__.__name__, ", ".join("%s=%s" % kv for kv in self._kwargs.items()))


class MaxPool2D(_Pooling):
    def __init__(self, pool_size=(2, 2), strides=None, padding=0, layout="NCHW", ceil_mode=False, **kwargs):
        super(MaxPool2D, self).__init__(pool_size, strides, padding, ceil_mode, False, "max", **kwargs)


class AvgPool2D(_Pooling):
    def __init__(self, pool_size=(2, 2), strides=None, padding=0, ceil_mode=False, layout="NCHW",
                 count_include_pad=True, **kwargs):
        super(AvgPool2D, self).__init__(pool_size, strides, padding, ceil_mode, False, "avg", count_include_pad,
                                        **kwargs)


class GlobalAvgPool2D(_Pooling):
    def __init__(self, layout="NCHW", **kwargs):
        super(GlobalAvgPool2D, self).__init__((1, 1), None, 0, True, True, "avg", **kwargs)


class GlobalMaxPool2D(_Pooling):
    def __init__(self, layout="NCHW", **kwargs):
        super(GlobalMaxPool2D, self).__init__((1, 1), None, 0, True, True, "max", **kwargs)


class Flatten(HybridBlock):
    def hybrid_forward(self, F, x):
        return F.Flatten(x)

    def __repr__(self):
        return "Flatten"


class Dropout(HybridBlock):
    def __init__(self, rate, axes=(), **kwargs):
        super(Dropout, self).__init__(**kwargs)
        self._rate = rate

    def hybrid_forward(self, F, x):
        return x            # inference only

    def __repr__(self):
        return "Dropout(p = %s)" % self._rate


class HybridLambda(HybridBlock):
    def __init__(self, function, prefix=None):
        super(HybridLambda, self).__init__(prefix=prefix)
        self._func = function

    def hybrid_forward(self, F, x, *args):
        return self._func(F, x, *args)
