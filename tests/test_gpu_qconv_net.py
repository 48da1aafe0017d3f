"""The reference's quantized MobileNet (tests/models/quantized_mobilenet.py -> nn/quantized_mobilenet.py) on the MI355X:
every `nn.Conv2D(quantized=True)` call of a forward - plain and with nn/fuse.py's producer fusion - recomputed by the C++
twin of the block from the very tensors the library was given: bit for bit."""
import numpy as np
import pytest
import torch

from oracle import host as H

pytestmark = pytest.mark.gpu


def _randomise_bn(net, rng, ctx):
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.mx.gluon import nn

    def visit(b):
        if type(b) is nn.BatchNorm:
            c = b.gamma.shape[0]
            assert c > 0, "BatchNorm without a known width (give in_channels)"
            b.gamma.set_data(mx.nd.array((rng.random(c) + 0.5).astype(np.float32), ctx=ctx))
            b.beta.set_data(mx.nd.array((rng.standard_normal(c) * 0.3).astype(np.float32), ctx=ctx))
            b.running_mean.set_data(mx.nd.array((rng.standard_normal(c) * 0.2).astype(np.float32), ctx=ctx))
            b.running_var.set_data(mx.nd.array((rng.random(c) + 0.5).astype(np.float32), ctx=ctx))
    net.apply(visit)


def _build(kind, mult, classes, ctx, seed=7):
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.nn import quantized_mobilenet as QM
    np.random.seed(seed)
    net = (QM.MobileNet if kind == "v1" else QM.MobileNetV2)(mult, classes=classes)
    net.initialize(mx.init.Xavier(magnitude=2.0), ctx=ctx)
    _randomise_bn(net, np.random.default_rng(seed), ctx)
    return net


class Spy(object):
    """Every ops.qconv2d call of a forward with its tensors."""

    def __init__(self, ops):
        self.ops, self.calls, self.real = ops, [], ops.qconv2d

    def __enter__(self):
        def spy(x, w, wbuf, bias, strides, padding, groups, ws, **kw):
            out = self.real(x, w, wbuf, bias, strides, padding, groups, ws, **kw)
            y, stat = out if isinstance(out, tuple) else (out, None)
            np_ = lambda t: None if t is None else t.detach().cpu().numpy()
            self.calls.append(dict(x=np_(x), w=np_(w), b=np_(bias), strides=tuple(strides), padding=tuple(padding),
                                   groups=groups, y=np_(y), stat=np_(stat), in_stat=np_(kw.get("in_stat")),
                                   bn_scale=np_(kw.get("bn_scale")), bn_shift=np_(kw.get("bn_shift")),
                                   act=kw.get("act", "none"), input_dtype=kw.get("input_dtype", "uint8")))
            return out
        self.ops.qconv2d = spy
        return self

    def __exit__(self, *exc):
        self.ops.qconv2d = self.real
        return False


def _check_calls(calls):
    for i, c in enumerate(calls):
        act = None if c["act"] == "none" else c["act"]
        want = H.qconv2d_forward(c["x"], c["w"], c["b"], c["strides"], c["padding"], c["groups"],
                                 input_dtype=c["input_dtype"], act=act, bn_scale=c["bn_scale"], bn_shift=c["bn_shift"],
                                 want_stat=c["stat"] is not None)
        if c["stat"] is not None:
            want, wstat = want
            np.testing.assert_array_equal(c["stat"], wstat, "call %d: per-sample statistic" % i)
        np.testing.assert_array_equal(c["y"], want, "call %d (%s)" % (i, c["x"].shape))
        if c["in_stat"] is not None:              # the producer's statistic IS the input's per-sample maximum
            assert c["x"].min() >= 0
            np.testing.assert_array_equal(c["in_stat"], c["x"].reshape(c["x"].shape[0], -1).max(axis=1))


@pytest.mark.parametrize("kind,mult,hw,batch", [("v1", 1.0, 64, 3), ("v1", 0.5, 96, 2), ("v2", 1.0, 64, 2)],
                         ids=["mobilenet1.0", "mobilenet0.5", "mobilenetv2_1.0"])
def test_every_quantised_convolution_of_the_net_equals_the_oracle(gpu, kind, mult, hw, batch, monkeypatch):
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.nn import fuse as qfuse
    net = _build(kind, mult, 100, gpu)
    rng = np.random.default_rng(3)
    x = mx.nd.array(rng.standard_normal((batch, 3, hw, hw)).astype(np.float32), ctx=gpu)
    # (MobileNetV2's classifier is a float 1x1 convolution of the tensor library, whose solver choice - and summation order -
    # may differ from one forward to the next: ask for its deterministic algorithm while forwards are compared bit for bit)
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    with Spy(ops) as spy:
        plain = net(x).asnumpy()
    n_q = len(spy.calls)
    assert n_q == (26 if kind == "v1" else 52)
    assert all(c["in_stat"] is None and c["bn_scale"] is None for c in spy.calls)
    _check_calls(spy.calls)
    # producer fusion: BatchNorm / ReLU folded into the stores, ranges from the producers' statistics
    n_fused = qfuse.fuse_inference(net)
    assert n_fused == n_q + (1 if mult == 1.0 else 0)                  # + the float first convolution (3 -> 32 only)
    with Spy(ops) as spy:
        fused = net(x).asnumpy()
    assert len(spy.calls) == n_q
    # (every consumer of a ReLU / ReLU6 output; with 16 stem channels the first convolution stays with the tensor library
    # and leaves no statistic for the first depthwise layer)
    assert sum(c["in_stat"] is not None for c in spy.calls) >= (26 if (kind, mult) == ("v1", 1.0) else 25 if kind == "v1" else 30)
    assert all(c["bn_scale"] is not None and c["stat"] is not None for c in spy.calls)
    _check_calls(spy.calls)
    # the range from the statistic is the range of the range pass: identical logits with the statistic switched off
    monkeypatch.setenv("FQ_QCONV_NO_STAT", "1")
    with Spy(ops) as spy:
        no_stat = net(x).asnumpy()
    assert all(c["in_stat"] is None for c in spy.calls)
    np.testing.assert_array_equal(no_stat, fused)
    monkeypatch.delenv("FQ_QCONV_NO_STAT")
    # fused vs plain: same net up to the BatchNorm's arithmetic (folded scale / shift vs the library's formula)
    assert np.isfinite(plain).all() and np.isfinite(fused).all()
    assert np.abs(fused - plain).max() <= 0.1 * np.abs(plain).max() + 1e-3
    qfuse.unfuse(net)
    np.testing.assert_array_equal(net(x).asnumpy(), plain)


def test_a_recomputed_layer_invalidates_its_statistic(gpu):
    """A producer whose layer goes through the exact direct kernel (here: `_input_range` = (1, 2) on a padded depthwise
    layer - the padding zero is clipped to a non-zero code) keeps no per-sample statistic: it says so (stat[0] = -1) and its
    consumer takes the range from the tensor - the chain equals the oracle's chain."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.nn import Conv2D, fuse as qfuse
    rng = np.random.default_rng(8)
    seq = nn.HybridSequential()
    seq.add(Conv2D(32, 3, 1, 1, in_channels=32, groups=32, use_bias=False, quantized=True, input_dtype="uint8",
                   weight_dtype="int8"), nn.BatchNorm(in_channels=32), nn.Activation("relu"),
            Conv2D(64, 1, 1, 0, in_channels=32, use_bias=False, quantized=True, input_dtype="uint8", weight_dtype="int8"),
            nn.BatchNorm(in_channels=64), nn.Activation("relu"))
    seq.initialize(mx.init.Xavier(), ctx=gpu)
    _randomise_bn(seq, rng, gpu)
    seq[0]._input_range = (1.0, 2.0)
    assert qfuse.fuse_inference(seq) == 2
    x = mx.nd.array((rng.random((3, 32, 14, 14)) * 1.5 + 0.8).astype(np.float32), ctx=gpu)
    with Spy(ops) as spy:
        seq(x)
    a, b = spy.calls
    assert a["stat"][0] == -1.0 and b["in_stat"][0] == -1.0
    want_a = H.qconv2d_forward(a["x"], a["w"], None, (1, 1), (1, 1), 32, input_range=(1.0, 2.0), act="relu",
                               bn_scale=a["bn_scale"], bn_shift=a["bn_shift"])
    np.testing.assert_array_equal(a["y"], want_a)
    want_b = H.qconv2d_forward(want_a, b["w"], None, (1, 1), (0, 0), 1, act="relu", bn_scale=b["bn_scale"],
                               bn_shift=b["bn_shift"])
    np.testing.assert_array_equal(b["y"], want_b)


def test_an_unpadded_consumer_of_a_tensor_without_zeros_takes_its_true_minimum(gpu):
    """uint8 without padding: the range starts at min(x).  A ReLU6 producer behind a BatchNorm with a large positive shift
    writes no zero at all - the consumer's scan runs to the end and finds the true minimum (codes from L > 0)."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.nn import Conv2D, fuse as qfuse
    from quantization.mxnet_amd.nn.quantized_mobilenet import RELU6
    rng = np.random.default_rng(12)
    seq = nn.HybridSequential()
    seq.add(Conv2D(64, 1, 1, 0, in_channels=32, use_bias=False, quantized=True, input_dtype="uint8", weight_dtype="int8"),
            nn.BatchNorm(in_channels=64), RELU6(),
            Conv2D(96, 1, 1, 0, in_channels=64, use_bias=False, quantized=True, input_dtype="uint8", weight_dtype="int8"),
            nn.BatchNorm(in_channels=96), nn.Activation("relu"))
    seq.initialize(mx.init.Xavier(), ctx=gpu)
    _randomise_bn(seq, rng, gpu)
    seq[1].beta.set_data(mx.nd.array(np.full(64, 4.0, np.float32), ctx=gpu))       # everything lands in (0, 6]
    seq[1].gamma.set_data(mx.nd.array(np.full(64, 0.05, np.float32), ctx=gpu))
    assert qfuse.fuse_inference(seq) == 2
    x = mx.nd.array(np.maximum(rng.standard_normal((4, 32, 14, 14)), 0).astype(np.float32), ctx=gpu)
    with Spy(ops) as spy:
        seq(x)
    a, b = spy.calls
    assert b["in_stat"] is not None and b["x"].min() > 0.5
    _check_calls(spy.calls)


def test_quantized_mobilenet_against_the_float_net_and_the_simulated_one(gpu):
    """The reference's own check of this model (tests/test_quantized_conv.py:60-79, `test_quantized_mobilnet`): the net built from
    `nn.Conv2D(quantized=True)`, the zoo's float mobilenet1.0 and the SIMULATED-quantisation net (`convert_model` with the first
    convolution excluded + `qparams_init`) on the same parameters and the same image batch.  The reference prints the three top-20
    lists; here (random parameters, no model store) the three logit tensors are held against each other: both quantised nets
    stay close to the float one, and closer to each other's side of it than an unrelated net would - plain and fused."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.mx.gluon.model_zoo import get_model
    from quantization.mxnet_amd.nn import fuse as qfuse
    from quantization.mxnet_amd.quantize import convert
    from quantization.mxnet_amd.quantize.initialize import qparams_init
    my_net = _build("v1", 1.0, 1000, gpu, seed=11)

    def same_parameters(net):
        net.initialize(ctx=gpu)
        mine = list(my_net.collect_params().values())
        theirs = list(net.collect_params().values())
        assert len(mine) == len(theirs)
        for a, b in zip(mine, theirs):                                 # (same creation order, same shapes: a gluoncv file loads)
            assert a.shape == b.shape, (a.name, b.name)
            b.set_data(a.data())
        net.collect_params().reset_ctx(gpu)
        return net
    ref_net = same_parameters(get_model("mobilenet1.0", classes=1000))
    sim_net = same_parameters(get_model("mobilenet1.0", classes=1000))
    convert.convert_model(sim_net, exclude=[sim_net.features[0]])
    qparams_init(sim_net)
    sim_net.collect_params().reset_ctx(gpu)
    x = mx.nd.array(np.random.default_rng(5).random((4, 3, 224, 224)).astype(np.float32), ctx=gpu)      # nd.uniform, as there
    ref = ref_net(x).asnumpy()
    sim = sim_net(x).asnumpy()
    my = my_net(x).asnumpy()
    scale = np.abs(ref).max()
    assert np.isfinite(my).all() and scale > 0
    # 8-bit per-tensor / per-layer quantisation of 27 layers: a few per cent of the logits' range
    assert np.abs(my - ref).max() < 0.2 * scale and np.abs(sim - ref).max() < 0.2 * scale
    assert np.abs(my - sim).max() < 0.2 * scale
    top = lambda t: np.argsort(-t, axis=1)[:, :20]
    overlap = lambda a, b: np.mean([len(set(r) & set(s)) / 20.0 for r, s in zip(top(a), top(b))])
    assert overlap(my, ref) >= 0.6 and overlap(my, sim) >= 0.6, (overlap(my, ref), overlap(my, sim))
    qfuse.fuse_inference(my_net)
    fused = my_net(x).asnumpy()
    assert np.abs(fused - my).max() <= 0.1 * scale and overlap(fused, ref) >= 0.6
