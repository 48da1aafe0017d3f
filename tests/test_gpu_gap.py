"""GPU parity of the pooled producer (round 6): fq_pwconv_i8_gap against fq_pwconv_i8[_strided] followed by fq_global_avg_pool_stat -
the two launches it replaces - bit for bit (means, per-sample statistic, `current_input_max`), against its host twin, and at net
level (MobileNet1.0, ResNet-50: logits and thresholds with and without).
Reference: gluon model_zoo MobileNet `features`: ... Conv2D(1x1) [convert_conv2d.py:53-66,108], BatchNorm, Activation,
GlobalAvgPool2D, Flatten; BottleneckV1's `(body(x) + shortcut).relu()` followed by the net's GlobalAvgPool2D."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ops():
    from quantization.mxnet_amd import ops as _ops
    return _ops


def _eq(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    assert not bad.any(), "%s: %d/%d mismatches; first at %s: got %r want %r" % (
        what, int(bad.sum()), a.size, np.argwhere(bad)[0], a[bad][0], b[bad][0])


def N(t):
    return t.detach().cpu().numpy()


def _t(a, dev):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# (n, cin, cout, h, w, residual)
CASES = [
    (128, 1024, 1024, 7, 7, False),     # MobileNet1.0's last 1x1 at the benchmark's batch: 512 channels per workgroup
    (6, 1024, 1024, 7, 7, False),       # few samples: 256 channels per workgroup
    (5, 512, 512, 7, 7, False),         # MobileNet0.5
    (3, 2048, 512, 7, 7, False),        # K / 32 = 64
    (4, 512, 2048, 7, 7, True),         # ResNet-50's last closing 1x1, with its residual operand
    (3, 512, 1024, 8, 8, False),        # 64-pixel planes (256 x 256 images)
    (5, 320, 1280, 7, 7, False),        # MobileNetV2's last 1x1 (K / 32 = 10, five channel groups of 256)
]
MODES = ["online_u8_bn_relu", "online_s8_bn_none", "offline_u8_bias_relu6"]


@pytest.mark.parametrize("case", CASES, ids=["%dx%d->%d@%dx%d%s" % (c[:5] + ("+res" if c[5] else "",)) for c in CASES])
@pytest.mark.parametrize("mode", MODES)
def test_gap_producer_equals_the_two_launches_and_the_host_twin(dev, ops, case, mode):
    n, cin, cout, h, w, with_res = case
    rng = np.random.default_rng(cin + cout + n)
    signed = "s8" in mode
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    if not signed:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin)) * 0.05).astype(np.float32)
    res = (rng.standard_normal((n, cout, h, w)) * 2).astype(np.float32) if with_res else None
    if "bias" in mode:
        bn, bias = (None, None), (rng.standard_normal(cout) * 0.1).astype(np.float32)
    else:
        bn, bias = ((0.5 + rng.random(cout)).astype(np.float32) * np.where(rng.random(cout) < 0.1, -1, 1).astype(np.float32),
                    (rng.standard_normal(cout) * 0.3).astype(np.float32)), None
    act = "relu6" if "relu6" in mode else ("relu" if "relu" in mode else None)
    xt = _t(x, dev)
    flags = ops.act_flags(signed=signed)
    codes, scales, rowsum = ops.weight_codes(_t(wt, dev), 1, 8)
    xstat = ops.absmax_per_sample(xt)
    plan = dict(in_thr=torch.full((1,), 2.5, device=dev), in_stat=xstat) if "offline" in mode else dict(in_stat=xstat)
    kw = dict(width=8, flags=flags, bn_scale=_t(bn[0], dev), bn_shift=_t(bn[1], dev), act=act, residual=_t(res, dev))
    cur_a, cur_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    y, _ = ops.pwconv_i8(xt, codes, scales, rowsum, _t(bias, dev), cur_out=cur_a, **kw, **plan)
    want, want_stat = ops.global_avg_pool_stat(y, want_stat=True)
    got, got_stat = ops.pwconv_i8_gap(xt, codes, scales, rowsum, _t(bias, dev), cur_out=cur_b, **kw, **plan)
    assert tuple(got.shape) == (n, cout, 1, 1)
    _eq(N(got).reshape(n, cout), N(want).reshape(n, cout), "plane means")
    _eq(N(got_stat), N(want_stat), "per-sample statistic of the means")
    _eq(N(cur_a), N(cur_b), "current_input_max")
    if n <= 8:
        from oracle import host as H
        hy, _ = H.pwconv_i8(x, wt.reshape(cout, cin, 1, 1), 1, 8, in_max=2.5 if "offline" in mode else None,
                            in_stat=H.absmax_per_sample(x), signed=signed, bias=bias, bn_scale=bn[0], bn_shift=bn[1], act=act,
                            want_stat=True, residual=res)
        hg, hs = H.global_avg_pool(hy, want_stat=True)
        _eq(N(got).reshape(n, cout), np.asarray(hg).reshape(n, cout), "host twin: means")
        _eq(N(got_stat), hs, "host twin: statistic")


def test_gap_producer_refuses_what_it_is_not_built_for(dev, ops):
    assert ops.pwconv_gap_supported((4, 1024, 7, 7), 1024) and not ops.pwconv_gap_supported((4, 1024, 14, 14), 1024)
    assert ops.pwconv_gap_supported((4, 320, 7, 7), 1280) and not ops.pwconv_gap_supported((4, 1024, 7, 7), 1000)
    assert not ops.pwconv_gap_supported((4, 192, 7, 7), 1280)
    x = torch.zeros(2, 1024, 14, 14, device=dev)
    codes, scales, rowsum = ops.weight_codes(torch.ones(1024, 1024, device=dev), 1, 8)
    with pytest.raises(ValueError):
        ops.pwconv_i8_gap(x, codes, scales, rowsum, in_stat=ops.absmax_per_sample(x))


@pytest.mark.parametrize("model,kw", [("mobilenet1.0", dict()), ("resnet50_v1", dict(quant_type="channel")), ("mobilenet0.5", dict())])
def test_a_net_with_the_pooled_producer_equals_the_same_net_without(dev, ops, model, kw):
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build as build
    rng = np.random.default_rng(21)
    X = mx.nd.array(rng.standard_normal((6, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0))
    outs = {}
    for on in (False, True):
        net = build(model, 1000, mx.gpu(0), **kw)
        net.fix_params()
        net.quantize_input(enable=True, online=True)
        net(mx.nd.NDArray(X._t[:2].contiguous()))
        fuse.fuse_inference(net)
        old, fuse.GAP_FUSE = fuse.GAP_FUSE, on
        seen = []
        real = ops.pwconv_i8_gap
        ops.pwconv_i8_gap = lambda *a, **k: (seen.append(tuple(a[0].shape)), real(*a, **k))[1]
        try:
            out = net(X)
            cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
            net.update_ema()
            thr = np.asarray([b.input_max.data().asscalar() for b in net.collect_quantized_blocks()], np.float32)
        finally:
            fuse.GAP_FUSE = old
            ops.pwconv_i8_gap = real
        outs[on] = (N(out._t), cur, thr, seen)
    assert outs[False][3] == [] and len(outs[True][3]) == 1, outs[True][3]
    _eq(outs[True][0], outs[False][0], "logits")
    _eq(outs[True][1], outs[False][1], "current_input_max of every block")
    _eq(outs[True][2], outs[False][2], "thresholds after one naive-EMA step")
    # a hook on the pooling block keeps the two launches (it is shown the planes)
    net = build(model, 1000, mx.gpu(0), **kw)
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    net(mx.nd.NDArray(X._t[:2].contiguous()))
    fuse.fuse_inference(net)
    gap = [b for b in net.features._children.values() if type(b).__name__ == "GlobalAvgPool2D"][0]
    shapes = []
    hk = gap.register_forward_pre_hook(lambda blk, inp: shapes.append(tuple(inp[0].shape)))
    try:
        got = net(X).asnumpy()
    finally:
        hk.detach()
    assert len(shapes) == 1 and shapes[0][2:] == (7, 7)
    _eq(got, outs[True][0], "logits with a hook on the pooling block")


def test_mobilenetv2_offline_with_the_pooled_producer_equals_the_same_net_without(dev, ops):
    """BASELINE configuration 4 (per-channel W4A8, stored thresholds): the last 1x1 (320 -> 1280, ReLU6) hands the plane means to the
    un-quantised classifier convolution; logits and every block's current_input_max bit-equal with and without."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build as build
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        rng = np.random.default_rng(4)
        xs = [mx.nd.array(rng.standard_normal((6, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0)) for _ in range(3)]
        outs = {}
        for on in (False, True):
            net = build("mobilenetv2_1.0", 1000, mx.gpu(0), quant_type="channel", wt=4)
            net.quantize_input(enable=True, online=True)
            for x in xs[:2]:
                net(x)
                net.update_ema()
            net.fix_params()
            net.quantize_input(enable=True, online=False)
            net(xs[2])
            fuse.fuse_inference(net)
            old, fuse.GAP_FUSE = fuse.GAP_FUSE, on
            seen = []
            real = ops.pwconv_i8_gap
            ops.pwconv_i8_gap = lambda *a, **k: (seen.append(tuple(a[0].shape)), real(*a, **k))[1]
            try:
                out = net(xs[2])
                cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
            finally:
                fuse.GAP_FUSE = old
                ops.pwconv_i8_gap = real
            outs[on] = (N(out._t), cur, seen)
        assert outs[False][2] == [] and outs[True][2] == [(6, 320, 7, 7)], outs[True][2]
        _eq(outs[True][0], outs[False][0], "logits")
        _eq(outs[True][1], outs[False][1], "current_input_max of every block")
    finally:
        torch.backends.cudnn.deterministic = was
