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

__all__ = ['MobileNet', 'MobileNetV2', 'mobilenet1_0', 'mobilenet_v2_1_0', 'mobilenet0_75', 'mobilenet_v2_0_75',
           'mobilenet0_5', 'mobilenet_v2_0_5', 'mobilenet0_25', 'mobilenet_v2_0_25', 'get_mobilenet', 'get_mobilenet_v2']


class RELU6(HybridBlock):
    """Relu6 used in MobileNetV2 (:47-54)."""

    def hybrid_forward(self, F, x):
        return F.clip(x, 0, 6)


def _add_conv(out, channels=1, kernel=1, stride=1, pad=0, in_channels=3, num_group=1, active=True, relu6=False,
              quantized=True):
    """(:57-67)"""
    if quantized:
        out.add(QConv2D(channels, kernel, stride, pad, in_channels=in_channels, groups=num_group, use_bias=False,
                        quantized=True, input_dtype="uint8", weight_dtype="int8"))
    else:
        out.add(QConv2D(channels, kernel, stride, pad, in_channels=in_channels, groups=num_group, use_bias=False))
    out.add(nn.BatchNorm(scale=True, in_channels=channels))     # (the reference leaves the width to deferred initialisation)
    if active:
        out.add(RELU6() if relu6 else nn.Activation('relu'))


def _add_conv_dw(out, dw_channels, channels, stride, relu6=False):
    """(:70-76)"""
    _add_conv(out, channels=dw_channels, kernel=3, stride=stride, in_channels=dw_channels, pad=1, num_group=dw_channels,
              relu6=relu6)
    _add_conv(out, channels=channels, relu6=relu6, in_channels=dw_channels)


class LinearBottleneck(HybridBlock):
    """(:79-134)"""

    def __init__(self, in_channels, channels, t, stride, **kwargs):
        super(LinearBottleneck, self).__init__(**kwargs)
        self.use_shortcut = stride == 1 and in_channels == channels
        with self.name_scope():
            self.out = nn.HybridSequential()
            _add_conv(self.out, in_channels * t, relu6=True, in_channels=in_channels)
            _add_conv(self.out, in_channels * t, kernel=3, stride=stride, pad=1, num_group=in_channels * t, relu6=True,
                      in_channels=in_channels * t)
            _add_conv(self.out, channels, active=False, relu6=True, in_channels=in_channels * t)

    def forward(self, x):
        out = self.out(x)
        if self.use_shortcut:
            out = out + x
        return out

    def hybrid_forward(self, F, x):
        return self.forward(x)


class MobileNet(HybridBlock):
    """(:137-185)"""

    def __init__(self, multiplier=1.0, classes=1000, **kwargs):
        super(MobileNet, self).__init__(**kwargs)
        with self.name_scope():
            self.features = nn.HybridSequential(prefix='')
            with self.features.name_scope():
                _add_conv(self.features, channels=int(32 * multiplier), kernel=3, pad=1, stride=2, in_channels=3,
                          quantized=False)
                dw_channels = [int(x * multiplier) for x in [32, 64] + [128] * 2 + [256] * 2 + [512] * 6 + [1024]]
                channels = [int(x * multiplier) for x in [64] + [128] * 2 + [256] * 2 + [512] * 6 + [1024] * 2]
                strides = [1, 2] * 3 + [1] * 5 + [2, 1]
                for dwc, c, s in zip(dw_channels, channels, strides):
                    _add_conv_dw(self.features, dw_channels=dwc, channels=c, stride=s)
                self.features.add(nn.GlobalAvgPool2D())
                self.features.add(nn.Flatten())
            self.output = nn.Dense(classes, in_units=channels[-1])

    def forward(self, x):
        return self.output(self.features(x))

    def hybrid_forward(self, F, x):
        return self.forward(x)


class MobileNetV2(HybridBlock):
    """(:188-250; see the module docstring for what differs)"""

    def __init__(self, multiplier=1.0, classes=1000, **kwargs):
        super(MobileNetV2, self).__init__(**kwargs)
        with self.name_scope():
            self.features = nn.HybridSequential(prefix='features_')
            with self.features.name_scope():
                _add_conv(self.features, int(32 * multiplier), kernel=3, stride=2, pad=1, relu6=True, in_channels=3,
                          quantized=False)
                in_channels_group = [int(x * multiplier) for x in [32] + [16] + [24] * 2 + [32] * 3 + [64] * 4 + [96] * 3
                                     + [160] * 3]
                channels_group = [int(x * multiplier) for x in [16] + [24] * 2 + [32] * 3 + [64] * 4 + [96] * 3 + [160] * 3
                                  + [320]]
                ts = [1] + [6] * 16
                strides = [1, 2] * 2 + [1, 1, 2] + [1] * 6 + [2] + [1] * 3
                for in_c, c, t, s in zip(in_channels_group, channels_group, ts, strides):
                    self.features.add(LinearBottleneck(in_channels=in_c, channels=c, t=t, stride=s))
                last_channels = int(1280 * multiplier) if multiplier > 1.0 else 1280
                _add_conv(self.features, last_channels, relu6=True, in_channels=channels_group[-1])
                self.features.add(nn.GlobalAvgPool2D())
            self.output = nn.HybridSequential(prefix='output_')
            with self.output.name_scope():
                self.output.add(QConv2D(classes, 1, 1, 0, in_channels=last_channels, use_bias=False, prefix='pred_'),
                                nn.Flatten())

    def forward(self, x):
        return self.output(self.features(x))

    def hybrid_forward(self, F, x):
        return self.forward(x)


def get_mobilenet(multiplier, pretrained=False, ctx=None, root='~/.mxnet/models', **kwargs):
    """(:253-293) `pretrained` needs gluoncv's model store, which this image does not have: load a parameter file with
    `net.load_parameters(path)` instead."""
    if pretrained:
        raise RuntimeError("pretrained weights come from gluoncv's model store (not available here): build the net and "
                           "call net.load_parameters(<mobilenet%s parameter file>)" % multiplier)
    return MobileNet(multiplier, **kwargs)


def get_mobilenet_v2(multiplier, pretrained=False, ctx=None, root='~/.mxnet/models', **kwargs):
    """(:296-340)"""
    if pretrained:
        raise RuntimeError("pretrained weights come from gluoncv's model store (not available here)")
    return MobileNetV2(multiplier, **kwargs)


def mobilenet1_0(**kwargs):
    return get_mobilenet(1.0, **kwargs)


def mobilenet_v2_1_0(**kwargs):
    return get_mobilenet_v2(1.0, **kwargs)


def mobilenet0_75(**kwargs):
    return get_mobilenet(0.75, **kwargs)


def mobilenet_v2_0_75(**kwargs):
    return get_mobilenet_v2(0.75, **kwargs)


def mobilenet0_5(**kwargs):
    return get_mobilenet(0.5, **kwargs)


def mobilenet_v2_0_5(**kwargs):
    return get_mobilenet_v2(0.5, **kwargs)


def mobilenet0_25(**kwargs):
    return get_mobilenet(0.25, **kwargs)


def mobilenet_v2_0_25(**kwargs):
    return get_mobilenet_v2(0.25, **kwargs)
