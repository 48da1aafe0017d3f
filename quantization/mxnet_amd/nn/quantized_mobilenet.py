"""MobileNet built from `nn.Conv2D(quantized=True)` blocks — the consumer of the stand-alone quantised convolution that
the reference keeps under tests/models/quantized_mobilenet.py (its `test_quantized_mobilnet`,
tests/test_quantized_conv.py:60-79, compares it with the zoo's mobilenet1.0 and with the simulated-quantisation net).

Same structure and constructor names as the reference file (:137-185, :252-330): a float first convolution
(`QConv2D(..., quantized=False)`), thirteen depthwise-separable pairs whose 3x3 depthwise and 1x1 pointwise convolutions are
`QConv2D(..., use_bias=False, quantized=True, input_dtype="uint8", weight_dtype="int8")`, each followed by BatchNorm and ReLU
as separate blocks (:57-67), global average pooling and a float Dense classifier.  Parameter names follow the reference's
prefixes, so a gluoncv mobilenet1.0 parameter file loads into it (`load_parameters`) as it does there.

`MobileNetV2` mirrors the reference's class of the same name (:188-250) with the two slips that keep the reference's version
from being constructed put right (its `_add_conv` calls omit `in_channels`, and its classifier passes `QConv2D` too few
arguments): every convolution gets its true input width and the classifier is a float 1x1 `QConv2D`.
"""
from ..mx.gluon import nn
from ..mx.gluon.block import HybridBlock
from .quantized_conv import Conv2D as QConv2D

__all__ = ['MobileNet', 'MobileNetV2', 'get_mobilenet', 'get_mobilenet_v2']

# (depthwise width, pointwise width, stride) of MobileNet's thirteen separable pairs at multiplier 1 (:150-153)
_V1_PAIRS = ((32, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 512, 2)) + \
            ((512, 512, 1),) * 5 + ((512, 1024, 2), (1024, 1024, 1))
# (input width, output width, expansion, stride) of MobileNetV2's seventeen bottlenecks at multiplier 1 (:207-214)
_V2_UNITS = ((32, 16, 1, 1), (16, 24, 6, 2), (24, 24, 6, 1), (24, 32, 6, 2), (32, 32, 6, 1), (32, 32, 6, 1), (32, 64, 6, 2)) + \
            ((64, 64, 6, 1),) * 3 + ((64, 96, 6, 1), (96, 96, 6, 1), (96, 96, 6, 1), (96, 160, 6, 2), (160, 160, 6, 1),
                                     (160, 160, 6, 1), (160, 320, 6, 1))


class RELU6(HybridBlock):
    """clip(x, 0, 6) as a block of its own (:47-54); nn/fuse.py recognises it by name."""

    def hybrid_forward(self, F, x):
        return F.clip(x, 0, 6)


def _conv_bn(seq, cin, cout, kernel=1, stride=1, groups=1, act="relu", quantized=True):
    """One `conv -> BatchNorm [-> activation]` triple appended to `seq`, blocks created in the reference's order (:57-67) so
    that the parameter names come out as they do there.  act: "relu", "relu6" or None."""
    extra = dict(quantized=True, input_dtype="uint8", weight_dtype="int8") if quantized else {}
    seq.add(QConv2D(cout, kernel, stride, kernel // 2, in_channels=cin, groups=groups, use_bias=False, **extra))
    seq.add(nn.BatchNorm(scale=True, in_channels=cout))          # (the reference leaves the width to deferred initialisation)
    if act == "relu6":
        seq.add(RELU6())
    elif act == "relu":
        seq.add(nn.Activation(act))


class LinearBottleneck(HybridBlock):
    """expansion 1x1 -> depthwise 3x3 -> linear projection 1x1, identity shortcut when the shape allows (:79-134)"""

    def __init__(self, in_channels, channels, t, stride, **kwargs):
        super().__init__(**kwargs)
        self.use_shortcut = (stride, in_channels) == (1, channels)
        wide = in_channels * t
        with self.name_scope():
            self.out = nn.HybridSequential()
            _conv_bn(self.out, in_channels, wide, act="relu6")
            _conv_bn(self.out, wide, wide, kernel=3, stride=stride, groups=wide, act="relu6")
            _conv_bn(self.out, wide, channels, act=None)

    def forward(self, x):
        y = self.out(x)
        return y + x if self.use_shortcut else y

    def hybrid_forward(self, F, x):
        return self.forward(x)


class MobileNet(HybridBlock):
    """(:137-185)"""

    def __init__(self, multiplier=1.0, classes=1000, **kwargs):
        super().__init__(**kwargs)
        scaled = lambda c: int(c * multiplier)
        with self.name_scope():
            self.features = nn.HybridSequential(prefix='')
            with self.features.name_scope():
                _conv_bn(self.features, 3, scaled(32), kernel=3, stride=2, quantized=False)
                for dw, pw, stride in _V1_PAIRS:
                    _conv_bn(self.features, scaled(dw), scaled(dw), kernel=3, stride=stride, groups=scaled(dw))
                    _conv_bn(self.features, scaled(dw), scaled(pw))
                self.features.add(nn.GlobalAvgPool2D(), nn.Flatten())
            self.output = nn.Dense(classes, in_units=scaled(_V1_PAIRS[-1][1]))

    def forward(self, x):
        return self.output(self.features(x))

    def hybrid_forward(self, F, x):
        return self.forward(x)


class MobileNetV2(HybridBlock):
    """(:188-250; see the module docstring for what differs)"""

    def __init__(self, multiplier=1.0, classes=1000, **kwargs):
        super().__init__(**kwargs)
        scaled = lambda c: int(c * multiplier)
        head = scaled(1280) if multiplier > 1.0 else 1280
        with self.name_scope():
            self.features = nn.HybridSequential(prefix='features_')
            with self.features.name_scope():
                _conv_bn(self.features, 3, scaled(32), kernel=3, stride=2, act="relu6", quantized=False)
                for cin, cout, t, stride in _V2_UNITS:
                    self.features.add(LinearBottleneck(in_channels=scaled(cin), channels=scaled(cout), t=t, stride=stride))
                _conv_bn(self.features, scaled(_V2_UNITS[-1][1]), head, act="relu6")
                self.features.add(nn.GlobalAvgPool2D())
            self.output = nn.HybridSequential(prefix='output_')
            with self.output.name_scope():
                self.output.add(QConv2D(classes, 1, 1, 0, in_channels=head, use_bias=False, prefix='pred_'), nn.Flatten())

    def forward(self, x):
        return self.output(self.features(x))

    def hybrid_forward(self, F, x):
        return self.forward(x)


def _no_model_store(what):
    raise RuntimeError("pretrained weights come from gluoncv's model store (not available here): build the net and call "
                       "net.load_parameters(<%s parameter file>)" % what)


def get_mobilenet(multiplier, pretrained=False, ctx=None, root='~/.mxnet/models', **kwargs):
    """(:253-293)"""
    if pretrained:
        _no_model_store("mobilenet%s" % multiplier)
    return MobileNet(multiplier, **kwargs)


def get_mobilenet_v2(multiplier, pretrained=False, ctx=None, root='~/.mxnet/models', **kwargs):
    """(:296-340)"""
    if pretrained:
        _no_model_store("mobilenetv2_%s" % multiplier)
    return MobileNetV2(multiplier, **kwargs)


def _factory(getter, multiplier, name):
    def make(**kwargs):
        return getter(multiplier, **kwargs)
    make.__name__ = make.__qualname__ = name
    make.__doc__ = "%s with width multiplier %s (the reference's constructor of the same name)" % (getter.__name__, multiplier)
    return make


# mobilenet1_0 ... mobilenet0_25, mobilenet_v2_1_0 ... mobilenet_v2_0_25 (:342-508)
for _m, _tag in ((1.0, "1_0"), (0.75, "0_75"), (0.5, "0_5"), (0.25, "0_25")):
    for _getter, _stem in ((get_mobilenet, "mobilenet"), (get_mobilenet_v2, "mobilenet_v2_")):
        _name = _stem + _tag
        globals()[_name] = _factory(_getter, _m, _name)
        __all__.append(_name)
