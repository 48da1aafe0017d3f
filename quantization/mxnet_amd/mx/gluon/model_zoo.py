"""Self-contained declarations of the four model families BASELINE.json's configs name, laid out exactly like
gluoncv's (`net.features[i]`, `net.output`, `features[2][0].body[0]` ...) because the reference CLI excludes blocks
by those positions (examples/simulate_quantization.py:238-244).  The layer lists follow the standard architectures
(MobileNet: cf. reference tests/models/quantized_mobilenet.py:137-185, which mirrors gluoncv's mobilenet1.0).

gluoncv's model zoo needs network access for `pretrained=True`; here weights are He-normal drawn from numpy's
global RNG (seed it with `np.random.seed(7)`, the CLI default) unless a parameter file is given.
"""
import os

from .block import HybridBlock
from . import nn
from .. import initializer as _init

__all__ = ["get_model", "get_model_list", "MobileNet", "MobileNetV2", "ResNetV1", "CIFARResNetV1", "VGG"]


class RELU6(HybridBlock):
    def hybrid_forward(self, F, x):
        return F.clip(x, 0, 6)

    def __repr__(self):
        return "RELU6"


def _add_conv(out, channels=1, kernel=1, stride=1, pad=0, num_group=1, active=True, relu6=False, in_channels=0):
    out.add(nn.Conv2D(channels, kernel, stride, pad, groups=num_group, use_bias=False, in_channels=in_channels))
    out.add(nn.BatchNorm(scale=True, in_channels=channels))
    if active:
        out.add(RELU6() if relu6 else nn.Activation("relu"))


def _add_conv_dw(out, dw_channels, channels, stride, relu6=False):
    _add_conv(out, channels=dw_channels, kernel=3, stride=stride, pad=1, num_group=dw_channels, relu6=relu6,
              in_channels=dw_channels)
    _add_conv(out, channels=channels, relu6=relu6, in_channels=dw_channels)


class MobileNet(HybridBlock):
    """features = [conv, bn, relu] + 13 x ([dw3x3, bn, relu], [pw1x1, bn, relu]) + [GlobalAvgPool2D, Flatten]."""

    def __init__(self, multiplier=1.0, classes=1000, **kwargs):
        super(MobileNet, self).__init__(**kwargs)
        with self.name_scope():
            self.features = nn.HybridSequential(prefix="")
            with self.features.name_scope():
                _add_conv(self.features, channels=int(32 * multiplier), kernel=3, pad=1, stride=2, in_channels=3)
                dw_channels = [int(x * multiplier) for x in [32, 64] + [128] * 2 + [256] * 2 + [512] * 6 + [1024]]
                channels = [int(x * multiplier) for x in [64] + [128] * 2 + [256] * 2 + [512] * 6 + [1024] * 2]
                strides = [1, 2] * 3 + [1] * 5 + [2, 1]
                for dwc, c, s in zip(dw_channels, channels, strides):
                    _add_conv_dw(self.features, dw_channels=dwc, channels=c, stride=s)
                self.features.add(nn.GlobalAvgPool2D())
                self.features.add(nn.Flatten())
            self.output = nn.Dense(classes, in_units=channels[-1])

    def hybrid_forward(self, F, x):
        return self.output(self.features(x))

    def forward(self, x):
        return self.output(self.features(x))


class LinearBottleneck(HybridBlock):
    def __init__(self, in_channels, channels, t, stride, **kwargs):
        super(LinearBottleneck, self).__init__(**kwargs)
        self.use_shortcut = stride == 1 and in_channels == channels
        with self.name_scope():
            self.out = nn.HybridSequential()
            _add_conv(self.out, in_channels * t, relu6=True, in_channels=in_channels)
            _add_conv(self.out, in_channels * t, kernel=3, stride=stride, pad=1, num_group=in_channels * t,
                      relu6=True, in_channels=in_channels * t)
            _add_conv(self.out, channels, active=False, relu6=True, in_channels=in_channels * t)

    def forward(self, x):
        out = self.out(x)
        if self.use_shortcut:
            out = out + x
        return out


class MobileNetV2(HybridBlock):
    def __init__(self, multiplier=1.0, classes=1000, **kwargs):
        super(MobileNetV2, self).__init__(**kwargs)
        with self.name_scope():
            self.features = nn.HybridSequential(prefix="features_")
            with self.features.name_scope():
                _add_conv(self.features, int(32 * multiplier), kernel=3, stride=2, pad=1, relu6=True, in_channels=3)
                in_channels_group = [int(x * multiplier) for x in
                                     [32] + [16] + [24] * 2 + [32] * 3 + [64] * 4 + [96] * 3 + [160] * 3]
                channels_group = [int(x * multiplier) for x in
                                  [16] + [24] * 2 + [32] * 3 + [64] * 4 + [96] * 3 + [160] * 3 + [320]]
                ts = [1] + [6] * 16
                strides = [1, 2] * 2 + [1, 1, 2] + [1] * 6 + [2] + [1] * 3
                for in_c, c, t, s in zip(in_channels_group, channels_group, ts, strides):
                    self.features.add(LinearBottleneck(in_channels=in_c, channels=c, t=t, stride=s))
                last_channels = int(1280 * multiplier) if multiplier > 1.0 else 1280
                _add_conv(self.features, last_channels, relu6=True, in_channels=channels_group[-1])
                self.features.add(nn.GlobalAvgPool2D())
            self.output = nn.HybridSequential(prefix="output_")
            with self.output.name_scope():
                self.output.add(nn.Conv2D(classes, 1, use_bias=False, prefix="pred_", in_channels=last_channels),
                                nn.Flatten())

    def forward(self, x):
        return self.output(self.features(x))


def _conv3x3(channels, stride, in_channels):
    return nn.Conv2D(channels, kernel_size=3, strides=stride, padding=1, use_bias=False, in_channels=in_channels)


class BottleneckV1(HybridBlock):
    def __init__(self, channels, stride, downsample=False, in_channels=0, **kwargs):
        super(BottleneckV1, self).__init__(**kwargs)
        self.body = nn.HybridSequential(prefix="")
        self.body.add(nn.Conv2D(channels // 4, kernel_size=1, strides=stride, in_channels=in_channels))
        self.body.add(nn.BatchNorm(in_channels=channels // 4))
        self.body.add(nn.Activation("relu"))
        self.body.add(_conv3x3(channels // 4, 1, channels // 4))
        self.body.add(nn.BatchNorm(in_channels=channels // 4))
        self.body.add(nn.Activation("relu"))
        self.body.add(nn.Conv2D(channels, kernel_size=1, strides=1, in_channels=channels // 4))
        self.body.add(nn.BatchNorm(in_channels=channels))
        if downsample:
            self.downsample = nn.HybridSequential(prefix="")
            self.downsample.add(nn.Conv2D(channels, kernel_size=1, strides=stride, use_bias=False,
                                          in_channels=in_channels))
            self.downsample.add(nn.BatchNorm(in_channels=channels))
        else:
            self.downsample = None

    def forward(self, x):
        residual = x
        x = self.body(x)
        if self.downsample is not None:
            residual = self.downsample(residual)
        return (x + residual).relu()


class BasicBlockV1(HybridBlock):
    def __init__(self, channels, stride, downsample=False, in_channels=0, **kwargs):
        super(BasicBlockV1, self).__init__(**kwargs)
        self.body = nn.HybridSequential(prefix="")
        self.body.add(_conv3x3(channels, stride, in_channels))
        self.body.add(nn.BatchNorm(in_channels=channels))
        self.body.add(nn.Activation("relu"))
        self.body.add(_conv3x3(channels, 1, channels))
        self.body.add(nn.BatchNorm(in_channels=channels))
        if downsample:
            self.downsample = nn.HybridSequential(prefix="")
            self.downsample.add(nn.Conv2D(channels, kernel_size=1, strides=stride, use_bias=False,
                                          in_channels=in_channels))
            self.downsample.add(nn.BatchNorm(in_channels=channels))
        else:
            self.downsample = None

    def forward(self, x):
        residual = x
        x = self.body(x)
        if self.downsample is not None:
            residual = self.downsample(residual)
        return (x + residual).relu()


class ResNetV1(HybridBlock):
    def __init__(self, block, layers, channels, classes=1000, thumbnail=False, **kwargs):
        super(ResNetV1, self).__init__(**kwargs)
        assert len(layers) == len(channels) - 1
        with self.name_scope():
            self.features = nn.HybridSequential(prefix="")
            if thumbnail:
                self.features.add(_conv3x3(channels[0], 1, 3))
            else:
                self.features.add(nn.Conv2D(channels[0], 7, 2, 3, use_bias=False, in_channels=3))
                self.features.add(nn.BatchNorm(in_channels=channels[0]))
                self.features.add(nn.Activation("relu"))
                self.features.add(nn.MaxPool2D(3, 2, 1))
            for i, num_layer in enumerate(layers):
                stride = 1 if i == 0 else 2
                self.features.add(self._make_layer(block, num_layer, channels[i + 1], stride, i + 1,
                                                   in_channels=channels[i]))
            self.features.add(nn.GlobalAvgPool2D())
            self.output = nn.Dense(classes, in_units=channels[-1])

    def _make_layer(self, block, layers, channels, stride, stage_index, in_channels=0):
        layer = nn.HybridSequential(prefix="stage%d_" % stage_index)
        with layer.name_scope():
            layer.add(block(channels, stride, channels != in_channels, in_channels=in_channels, prefix=""))
            for _ in range(layers - 1):
                layer.add(block(channels, 1, False, in_channels=channels, prefix=""))
        return layer

    def forward(self, x):
        return self.output(self.features(x))


class CIFARBasicBlockV1(BasicBlockV1):
    pass


class CIFARResNetV1(HybridBlock):
    def __init__(self, block, layers, channels, classes=10, **kwargs):
        super(CIFARResNetV1, self).__init__(**kwargs)
        assert len(layers) == len(channels) - 1
        with self.name_scope():
            self.features = nn.HybridSequential(prefix="")
            self.features.add(nn.Conv2D(channels[0], 3, 1, 1, use_bias=False, in_channels=3))
            self.features.add(nn.BatchNorm(in_channels=channels[0]))
            for i, num_layer in enumerate(layers):
                stride = 1 if i == 0 else 2
                self.features.add(self._make_layer(block, num_layer, channels[i + 1], stride, i + 1,
                                                   in_channels=channels[i]))
            self.features.add(nn.GlobalAvgPool2D())
            self.output = nn.Dense(classes, in_units=channels[-1])

    _make_layer = ResNetV1._make_layer

    def forward(self, x):
        return self.output(self.features(x))


class VGG(HybridBlock):
    """mxnet.gluon.model_zoo.vision.vgg (the net of the reference's tests/test_collect_qparams.py:14): features = stacks of
    [Conv2D 3x3 pad 1 (+ bias) (, BatchNorm), ReLU] each closed by MaxPool2D(2, 2), then Dense(4096, relu), Dropout, Dense(4096,
    relu), Dropout; output = Dense(classes)."""

    def __init__(self, layers, filters, classes=1000, batch_norm=False, **kwargs):
        super(VGG, self).__init__(**kwargs)
        assert len(layers) == len(filters)
        with self.name_scope():
            # (as gluon's vgg.py: every layer is created in THIS block's name scope - vgg0_conv0 ... vgg0_dense2)
            self.features = nn.HybridSequential(prefix="")
            cin = 3
            for num, ch in zip(layers, filters):
                for _ in range(num):
                    self.features.add(nn.Conv2D(ch, kernel_size=3, padding=1, in_channels=cin))
                    if batch_norm:
                        self.features.add(nn.BatchNorm(in_channels=ch))
                    self.features.add(nn.Activation("relu"))
                    cin = ch
                self.features.add(nn.MaxPool2D(strides=2))
            self.features.add(nn.Dense(4096, activation="relu"))
            self.features.add(nn.Dropout(rate=0.5))
            self.features.add(nn.Dense(4096, activation="relu", in_units=4096))
            self.features.add(nn.Dropout(rate=0.5))
            self.output = nn.Dense(classes, in_units=4096)

    def hybrid_forward(self, F, x):
        return self.output(self.features(x))


_VGG_SPEC = {11: ([1, 1, 2, 2, 2], [64, 128, 256, 512, 512]), 13: ([2, 2, 2, 2, 2], [64, 128, 256, 512, 512]),
             16: ([2, 2, 3, 3, 3], [64, 128, 256, 512, 512]), 19: ([2, 2, 4, 4, 4], [64, 128, 256, 512, 512])}


def _cifar_resnet(num_layers, **kw):
    assert (num_layers - 2) % 6 == 0
    n = (num_layers - 2) // 6
    return CIFARResNetV1(CIFARBasicBlockV1, [n] * 3, [16, 16, 32, 64], **kw)


_RESNET_SPEC = {18: (BasicBlockV1, [2, 2, 2, 2], [64, 64, 128, 256, 512]),
                34: (BasicBlockV1, [3, 4, 6, 3], [64, 64, 128, 256, 512]),
                50: (BottleneckV1, [3, 4, 6, 3], [64, 256, 512, 1024, 2048]),
                101: (BottleneckV1, [3, 4, 23, 3], [64, 256, 512, 1024, 2048]),
                152: (BottleneckV1, [3, 8, 36, 3], [64, 256, 512, 1024, 2048])}

_MODELS = {
    "mobilenet1.0": lambda **kw: MobileNet(1.0, **kw),
    "mobilenet0.75": lambda **kw: MobileNet(0.75, **kw),
    "mobilenet0.5": lambda **kw: MobileNet(0.5, **kw),
    "mobilenet0.25": lambda **kw: MobileNet(0.25, **kw),
    "mobilenetv2_1.0": lambda **kw: MobileNetV2(1.0, **kw),
    "mobilenetv2_0.75": lambda **kw: MobileNetV2(0.75, **kw),
    "mobilenetv2_0.5": lambda **kw: MobileNetV2(0.5, **kw),
    "mobilenetv2_0.25": lambda **kw: MobileNetV2(0.25, **kw),
    "cifar_resnet20_v1": lambda **kw: _cifar_resnet(20, **kw),
    "cifar_resnet56_v1": lambda **kw: _cifar_resnet(56, **kw),
    "cifar_resnet110_v1": lambda **kw: _cifar_resnet(110, **kw),
}
for _n, (_b, _l, _c) in _RESNET_SPEC.items():
    _MODELS["resnet%d_v1" % _n] = (lambda b, l, c: (lambda **kw: ResNetV1(b, l, c, **kw)))(_b, _l, _c)


for _n, (_l, _f) in _VGG_SPEC.items():
    _MODELS["vgg%d" % _n] = (lambda l, f: (lambda **kw: VGG(l, f, **kw)))(_l, _f)
    _MODELS["vgg%d_bn" % _n] = (lambda l, f: (lambda **kw: VGG(l, f, batch_norm=True, **kw)))(_l, _f)


def _models_dir(root=None):
    """`~/.mxnet/models` ($MXNET_HOME/models when that is set, as MXNet's `base.data_dir()` has it)."""
    return os.path.expanduser(root or os.path.join(os.environ.get("MXNET_HOME", os.path.join("~", ".mxnet")), "models"))


def find_checkpoint(name, root=None):
    """The parameter file `get_model(name, pretrained=True)` would use: gluoncv's `<name>-<hash>.params` (its
    `model_store.get_model_file` naming, MXNet NDArray-list format), a plain `<name>.params`, or this package's
    `<name>.params.npz` - the first that exists under `root` (default `~/.mxnet/models`); None when there is none."""
    import glob
    d = _models_dir(root)
    cands = sorted(glob.glob(os.path.join(d, glob.escape(name) + "-*.params"))) + \
        [os.path.join(d, name + ".params"), os.path.join(d, name + ".params.npz")]
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


def get_model_list():
    return list(_MODELS.keys())


def get_model(name, pretrained=False, classes=None, ctx=None, root=None, **kwargs):
    """gluoncv.model_zoo.get_model(name, pretrained=True, classes=...) (examples/simulate_quantization.py:188-204).

    `pretrained` may be a path to a parameter file (MXNet's `.params` NDArray-list format with gluoncv's structural names, or
    this package's npz); `True` looks under `<root or ~/.mxnet/models>` for gluoncv's `<name>-<hash>.params`, `<name>.params`
    or `<name>.params.npz` (`find_checkpoint`) and otherwise falls back to seeded He-normal weights with a notice (there is no
    network to fetch gluoncv's checkpoints).
    """
    name = name.lower()
    if name not in _MODELS:
        raise ValueError("Model %s is not supported. Available: %s" % (name, ", ".join(sorted(_MODELS))))
    from .block import reset_naming
    reset_naming()
    kw = {}
    if classes is not None:
        kw["classes"] = classes
    # gluoncv's constructor arguments the reference CLI can pass (examples/simulate_quantization.py:188-204): honoured where the
    # zoo has the feature, refused as gluoncv would refuse an argument the class does not take - never dropped silently
    last_gamma = bool(kwargs.pop("last_gamma", False))
    if kwargs.pop("use_se", False):
        raise NotImplementedError("use_se=True: the model zoo of this build has no squeeze-and-excitation variants")
    kwargs.pop("norm_layer", None)
    batch_norm = bool(kwargs.pop("batch_norm", False))    # (the reference CLI passes it to vgg only: `vgg16` + batch_norm = vgg16_bn)
    if batch_norm:
        if not name.startswith("vgg"):
            raise TypeError("%s: __init__() got an unexpected keyword argument 'batch_norm'" % name)
        if not name.endswith("_bn"):
            kw["batch_norm"] = True
    if last_gamma and not (name.startswith("resnet") or name.startswith("cifar_resnet")):
        raise TypeError("%s: __init__() got an unexpected keyword argument 'last_gamma'" % name)
    if kwargs:
        raise TypeError("%s: __init__() got unexpected keyword arguments %s" % (name, sorted(kwargs)))
    net = _MODELS[name](**kw)
    net.initialize(_init.MSRAPrelu(factor_type="in", slope=0.0), ctx=ctx)
    if last_gamma:
        # gluoncv resnet `last_gamma`: the last BatchNorm of every residual unit starts at gamma = 0 (the unit is the identity)
        def zero_last(b):
            body = getattr(b, "body", None)
            if isinstance(b, (BasicBlockV1, BottleneckV1, CIFARBasicBlockV1)) and body is not None:
                last = list(body._children.values())[-1]
                if type(last) is nn.BatchNorm:
                    last.gamma.set_data(last.gamma.data() * 0)
        net.apply(zero_last)
    path = None
    if isinstance(pretrained, str):
        path = pretrained
    elif pretrained:
        path = find_checkpoint(name, root)
        if path is None:
            print("[model_zoo] no checkpoint for %s under %s (no network): using seeded He-normal weights"
                  % (name, _models_dir(root)))
        else:
            print("[model_zoo] %s: parameters from %s" % (name, path))
    if path is not None:
        net.load_parameters(path, ctx=ctx)
    return net
