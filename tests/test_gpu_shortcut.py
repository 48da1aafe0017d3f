"""GPU parity of the folded shortcut (round 6): fq_pwconv_i8_shortcut against the two launches it replaces -
fq_pwconv_i8_strided(x2: the shortcut convolution + BatchNorm, no activation) then fq_pwconv_i8_strided(x, residual = that) - bit for
bit (output, per-sample statistic, both `current_input_max`), against its host twin, and at net level (ResNet-50 with and without).
Reference: gluon model_zoo BottleneckV1 with `downsample`: `(body(x) + downsample(x)).relu()`, both branches ending in a Conv2D(1x1)
wrapped by quantize/convert/convert_conv2d.py:53-66,108 and a BatchNorm."""
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


# (n, cin, cin2, cout, h, w)
CASES = [
    (3, 64, 64, 256, 56, 56),        # ResNet-50 stage 1
    (3, 128, 256, 512, 28, 28),      # stage 2 (the shortcut reads the subsampled trunk)
    (5, 256, 512, 1024, 14, 14),     # stage 3
    (9, 256, 512, 1024, 7, 7),       # tiles straddle samples (49 pixels), nine samples
    (6, 512, 1024, 2048, 7, 7),      # stage 4
    (2, 64, 64, 256, 5, 3),          # tiny planes: several samples per tile
]
MODES = ["online_u8_relu", "online_s8_none", "offline_u8_relu", "mixed_relu6_bias"]


def _make(case, mode):
    n, cin, cin2, cout, h, w = case
    rng = np.random.default_rng(cin + cin2 + h)
    signed = "s8" in mode
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32) * np.float32(1.4)
    x2 = rng.standard_normal((n, cin2, h, w)).astype(np.float32) * np.float32(2.1)
    if not signed:
        x, x2 = np.maximum(x, 0), np.maximum(x2, 0)
    k = dict(case=case, signed=signed, x=x, x2=x2, mode=mode)
    k["w"] = (rng.standard_normal((cout, cin)) * 0.1).astype(np.float32)
    k["w2"] = (rng.standard_normal((cout, cin2)) * 0.05).astype(np.float32)
    k["bn"] = ((0.5 + rng.random(cout)).astype(np.float32) * np.where(rng.random(cout) < 0.1, -1, 1).astype(np.float32),
               (rng.standard_normal(cout) * 0.3).astype(np.float32))
    k["bn2"] = ((0.5 + rng.random(cout)).astype(np.float32), (rng.standard_normal(cout) * 0.3).astype(np.float32))
    k["bias"] = (rng.standard_normal(cout) * 0.1).astype(np.float32) if "bias" in mode else None
    k["act"] = "relu6" if "relu6" in mode else ("relu" if "relu" in mode else None)
    return k


def _run(k, dev, ops, fused):
    n, cin, cin2, cout, h, w = k["case"]
    x, x2 = _t(k["x"], dev), _t(k["x2"], dev)
    flags = ops.act_flags(signed=k["signed"])
    c1 = ops.weight_codes(_t(k["w"], dev), 1, 8)
    c2 = ops.weight_codes(_t(k["w2"], dev), 1, 8)
    st1, st2 = ops.absmax_per_sample(x), ops.absmax_per_sample(x2)
    cur1, cur2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    offline1 = "offline" in k["mode"]
    offline2 = "offline" in k["mode"] or "mixed" in k["mode"]
    p1 = dict(in_thr=torch.full((1,), 3.0, device=dev), in_stat=st1) if offline1 else dict(in_stat=st1)
    p2 = dict(in_thr=torch.full((1,), 4.5, device=dev), in_stat=st2) if offline2 else dict(in_stat=st2)
    bn, bn2 = (_t(k["bn"][0], dev), _t(k["bn"][1], dev)), (_t(k["bn2"][0], dev), _t(k["bn2"][1], dev))
    if not fused:
        s, _ = ops.pwconv_i8(x2, *c2, None, width=8, flags=flags, cur_out=cur2, bn_scale=bn2[0], bn_shift=bn2[1], act=None,
                             want_stat=False, form="split", **p2)
        y, stat = ops.pwconv_i8(x, *c1, _t(k["bias"], dev), width=8, flags=flags, cur_out=cur1, bn_scale=bn[0], bn_shift=bn[1],
                                act=k["act"], residual=s, **p1)
    else:
        y, stat = ops.pwconv_i8_shortcut(x, *c1, _t(k["bias"], dev), width=8, flags=flags, cur_out=cur1, bn_scale=bn[0],
                                         bn_shift=bn[1], act=k["act"], x2=x2, wcodes2=c2[0], wscale2=c2[1], wsum2=c2[2],
                                         width2=8, flags2=flags, cur_out2=cur2, bn_scale2=bn2[0], bn_shift2=bn2[1],
                                         in_stat2=p2.get("in_stat"), in_thr2=p2.get("in_thr"), **p1)
    return N(y), N(stat), N(cur1), N(cur2)


@pytest.mark.parametrize("case", CASES, ids=["%dx%d+%d->%d@%dx%d" % c for c in CASES])
@pytest.mark.parametrize("mode", MODES)
def test_folded_shortcut_equals_the_two_launches_and_the_host_twin(dev, ops, case, mode):
    k = _make(case, mode)
    got = _run(k, dev, ops, True)
    want = _run(k, dev, ops, False)
    for a, b, what in zip(got, want, ("output", "statistic", "current_input_max of the closing convolution",
                                      "current_input_max of the shortcut convolution")):
        _eq(a, b, what)
    from oracle import host as H
    n, cin, cin2, cout, h, w = case
    off1 = "offline" in mode
    off2 = "offline" in mode or "mixed" in mode
    s = H.pwconv_i8(k["x2"], k["w2"].reshape(cout, cin2, 1, 1), 1, 8, in_max=4.5 if off2 else None,
                    in_stat=H.absmax_per_sample(k["x2"]), signed=k["signed"], bn_scale=k["bn2"][0], bn_shift=k["bn2"][1], act=None)
    hy, hstat = H.pwconv_i8(k["x"], k["w"].reshape(cout, cin, 1, 1), 1, 8, in_max=3.0 if off1 else None,
                            in_stat=H.absmax_per_sample(k["x"]), signed=k["signed"], bias=k["bias"], bn_scale=k["bn"][0],
                            bn_shift=k["bn"][1], act=k["act"], want_stat=True, residual=s)
    _eq(got[0], hy, "host twins: output")
    _eq(got[1], hstat, "host twins: statistic")


def test_folded_shortcut_refuses_what_it_is_not_built_for(dev, ops):
    assert ops.pwconv_shortcut_supported(64, 64, 256) and ops.pwconv_shortcut_supported(256, 512, 1024)
    assert ops.pwconv_shortcut_supported(512, 1024, 2048) and not ops.pwconv_shortcut_supported(512, 1024, 1280)
    assert not ops.pwconv_shortcut_supported(64, 128, 256) and not ops.pwconv_shortcut_supported(64, 64, 128)


@pytest.mark.parametrize("wino", ["none", "F43"])
def test_resnet50_with_folded_shortcuts_equals_the_same_net_without(dev, ops, wino):
    """Four launches compute their unit's shortcut convolution themselves (stage 1 on the pooled input, stages 2 to 4 on the
    subsampled trunk): logits, every block's current_input_max and the thresholds after a naive-EMA step are bit-equal."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build as build
    rng = np.random.default_rng(31)
    X = mx.nd.array(rng.standard_normal((4, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0))
    outs = {}
    for on in (False, True):
        net = build("resnet50_v1", 1000, mx.gpu(0), quant_type="channel", wino=wino)
        net.fix_params()
        net.quantize_input(enable=True, online=True)
        net(mx.nd.NDArray(X._t[:2].contiguous()))
        fuse.fuse_inference(net)
        old, fuse.SHORTCUT_FUSE = fuse.SHORTCUT_FUSE, on
        seen = []
        real = ops.pwconv_i8_shortcut
        ops.pwconv_i8_shortcut = lambda *a, **k: (seen.append((tuple(a[0].shape), tuple(k["x2"].shape))), real(*a, **k))[1]
        try:
            out = net(X)
            cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
            net.update_ema()
            thr = np.asarray([b.input_max.data().asscalar() for b in net.collect_quantized_blocks()], np.float32)
        finally:
            fuse.SHORTCUT_FUSE = old
            ops.pwconv_i8_shortcut = real
        outs[on] = (N(out._t), cur, thr, seen)
    assert outs[False][3] == [] and len(outs[True][3]) == 4, outs[True][3]
    assert outs[True][3][0] == ((4, 64, 56, 56), (4, 64, 56, 56)) and outs[True][3][1] == ((4, 128, 28, 28), (4, 256, 28, 28))
    _eq(outs[True][0], outs[False][0], "logits")
    _eq(outs[True][1], outs[False][1], "current_input_max of every block")
    _eq(outs[True][2], outs[False][2], "thresholds after one naive-EMA step")


def test_a_deferred_shortcut_nobody_folds_is_materialised(dev, ops):
    """With the subsampled trunk switched off the strided shortcut convolutions of stages 2-4 run as they are (stride 2: no record);
    and a unit whose closing convolution left the integer path computes the shortcut tensor after all - same logits."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build as build
    rng = np.random.default_rng(32)
    X = mx.nd.array(rng.standard_normal((2, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0))
    net = build("resnet50_v1", 1000, mx.gpu(0), quant_type="channel")
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    net(X)
    fuse.fuse_inference(net)
    want = net(X).asnumpy()
    seen = []
    real = ops.pwconv_i8_shortcut
    ops.pwconv_i8_shortcut = lambda *a, **k: (seen.append(tuple(a[0].shape)), real(*a, **k))[1]
    old = fuse.SUBSAMPLE
    try:
        fuse.SUBSAMPLE = False
        got = net(X).asnumpy()
        assert seen == [(2, 64, 56, 56)]                       # (only stage 1's shortcut has stride 1 on its own)
        _eq(got, want, "logits without the subsampled trunk")
        fuse.SUBSAMPLE = True
        # stage 1's closing convolution off the integer path: the record is materialised, the library convolution adds it
        unit = list(list(net.features._children.values())[4]._children.values())[0]
        tail = list(unit.body._children.values())[-2]
        tail._fq_no_int8 = True
        del seen[:]
        got = net(X).asnumpy()
        assert len(seen) == 3 and (2, 64, 56, 56) not in seen
        # (a library convolution of the fake-quantised tensors where the integer path was: another summation order, amplified by
        # the online thresholds behind it - the distance fusing itself has)
        assert np.abs(got - want).max() <= 2e-2 * np.abs(want).max()
    finally:
        fuse.SUBSAMPLE = old
        ops.pwconv_i8_shortcut = real


C16_CASES = [(3, 64, 64, 256, 56, 56, False), (3, 128, 256, 512, 28, 28, True), (3, 128, 256, 512, 28, 28, False),
             (5, 256, 512, 1024, 14, 14, True), (9, 256, 512, 1024, 7, 7, False), (4, 512, 1024, 2048, 7, 7, True)]


@pytest.mark.parametrize("case", C16_CASES, ids=["%dx%d+%d->%d@%dx%d%s" % (c[:6] + ("-codes" if c[6] else "-fp32",)) for c in C16_CASES])
def test_folded_shortcut_under_stored_thresholds_equals_the_two_launches(dev, ops, case):
    """fq_pwconv_i8_shortcut_c16: x as codes, y as fp32 + code copy, the shortcut's input fp32 or codes - against the shortcut's own
    launch followed by fq_pwconv_i8_c16_dual with it as residual: y, code copy, statistic, both current_input_max."""
    from oracle import fq_oracle as O
    n, cin, cin2, cout, h, w, b16 = case
    rng = np.random.default_rng(cin + cin2 + h + int(b16))
    x = np.maximum(rng.standard_normal((n, cin, h, w)) * 1.5, 0).astype(np.float32)
    x2 = np.maximum(rng.standard_normal((n, cin2, h, w)) * 2.0, 0).astype(np.float32)
    thr, thr2, thr3 = np.float32(2.3), np.float32(3.7), np.float32(4.1)
    thr_t, thr2_t, thr3_t = (_t(np.float32([v]), dev) for v in (thr, thr2, thr3))
    cx = O.ste_codes(x, O.act_scale(thr, False, 8), thr, np.float32(0))
    xc = ops.Codes16(_t(O.to_c16(cx.astype(np.int64), 128), dev), x.shape, thr_t, 8, 0)
    if b16:
        cx2 = O.ste_codes(x2, O.act_scale(thr2, False, 8), thr2, np.float32(0))
        x2a = ops.Codes16(_t(O.to_c16(cx2.astype(np.int64), 128), dev), x2.shape, thr2_t, 8, 0)
    else:
        x2a = _t(x2, dev)
    c1 = ops.weight_codes(_t((rng.standard_normal((cout, cin)) * 0.1).astype(np.float32), dev), 1, 8)
    c2 = ops.weight_codes(_t((rng.standard_normal((cout, cin2)) * 0.05).astype(np.float32), dev), 1, 8)
    bn = (_t((0.5 + rng.random(cout)).astype(np.float32), dev), _t((rng.standard_normal(cout) * 0.3).astype(np.float32), dev))
    bn2 = (_t((0.5 + rng.random(cout)).astype(np.float32), dev), _t((rng.standard_normal(cout) * 0.3).astype(np.float32), dev))
    st1, st2 = _t(O.absmax_per_sample(x), dev), _t(O.absmax_per_sample(x2), dev)
    side = dict(thr=thr3_t, width=8, flags=0)
    cur = [torch.zeros(1, device=dev) for _ in range(4)]
    s, _ = ops.pwconv_i8(x2a, *c2, None, in_thr=thr2_t, in_stat=st2, width=8, flags=0, cur_out=cur[0], bn_scale=bn2[0], bn_shift=bn2[1],
                         act=None, want_stat=False)
    want, want_stat, want16 = ops.pwconv_i8(xc, *c1, None, in_thr=thr_t, in_stat=st1, width=8, flags=0, cur_out=cur[1], bn_scale=bn[0],
                                            bn_shift=bn[1], act="relu", residual=s, side_codes=side)
    got, got_stat, got16 = ops.pwconv_i8_shortcut(xc, *c1, None, in_thr=thr_t, in_stat=st1, width=8, flags=0, cur_out=cur[3],
                                                  bn_scale=bn[0], bn_shift=bn[1], act="relu", x2=x2a, wcodes2=c2[0], wscale2=c2[1],
                                                  wsum2=c2[2], in_thr2=thr2_t, in_stat2=st2, width2=8, flags2=0, cur_out2=cur[2],
                                                  bn_scale2=bn2[0], bn_shift2=bn2[1], side_codes=side)
    _eq(N(got), N(want), "fp32 output")
    _eq(N(got16.t), N(want16.t), "code copy")
    _eq(N(got_stat), N(want_stat), "statistic")
    _eq(N(cur[2]), N(cur[0]), "current_input_max of the shortcut convolution")
    _eq(N(cur[3]), N(cur[1]), "current_input_max of the closing convolution")


def test_resnet50_offline_with_folded_shortcuts_equals_the_same_net_without(dev, ops):
    """BASELINE configuration 3's evaluation (stored thresholds, codes between the layers): the three stage heads fold their shortcut
    convolution (fq_pwconv_i8_shortcut_c16: codes in, fp32 + code copy out); logits and every current_input_max bit-equal."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        rng = np.random.default_rng(18)
        xs = [mx.nd.array(rng.standard_normal((4, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0)) for _ in range(3)]
        outs = {}
        for on in (False, True):
            net = _build("resnet50_v1", 1000, mx.gpu(0), quant_type="channel")
            net.quantize_input(enable=True, online=True)
            for x in xs[:2]:
                net(x)
                net.update_ema()
            net.fix_params()
            net.quantize_input(enable=True, online=False)
            net(xs[2])
            fuse.fuse_inference(net)
            old, fuse.SHORTCUT_FUSE = fuse.SHORTCUT_FUSE, on
            seen = []
            real = ops.pwconv_i8_shortcut

            def spy(*a, **k):
                seen.append((isinstance(a[0], ops.Codes16), isinstance(k["x2"], ops.Codes16), k.get("side_codes") is not None))
                return real(*a, **k)
            ops.pwconv_i8_shortcut = spy
            try:
                out = net(xs[2])
                cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
            finally:
                fuse.SHORTCUT_FUSE = old
                ops.pwconv_i8_shortcut = real
            outs[on] = (N(out._t), cur, seen)
        # stage 1: the pooled first convolution's output is fp32; stages 2 and 3: the subsampled trunk's code copy
        assert outs[False][2] == [] and outs[True][2] == [(True, False, True)] + [(True, True, True)] * 3, outs[True][2]
        _eq(outs[True][0], outs[False][0], "logits")
        _eq(outs[True][1], outs[False][1], "current_input_max of every block")
    finally:
        torch.backends.cudnn.deterministic = was
