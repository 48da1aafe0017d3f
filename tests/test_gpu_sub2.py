"""GPU parity of the subsampled trunk (round 6): fq_pwconv_i8_sub2 against
  (a) fq_pwconv_i8_strided(stride = 1, residual) - the launch it replaces - of which it must store exactly y[:, :, ::2, ::2], with
      the same per-sample statistic (taken over ALL of y) and the same `current_input_max`, bit for bit;
  (b) its host twin (oracle/fq_host.cpp);
and, at net level, ResNet-50 v1 with the three stage boundaries subsampled against the same net without: logits, every block's
`current_input_max` and the thresholds after a naive-EMA step bit-equal.
Reference: gluon model_zoo BottleneckV1 - `(body(x) + downsample(x)).relu()` whose output feeds the next stage's `body[0]` and
`downsample[0]`, both Conv2D(1x1, stride 2) wrapped by quantize/convert/convert_conv2d.py:53-66,108."""
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


# (n, cin, cout, h, w)
CASES = [
    (3, 64, 256, 56, 56),        # ResNet-50 stage 1 -> 2
    (3, 128, 512, 28, 28),       # stage 2 -> 3
    (5, 256, 1024, 14, 14),      # stage 3 -> 4
    (2, 512, 2048, 7, 7),        # odd planes: ceil(7 / 2) = 4 stored rows and columns
    (3, 64, 160, 9, 13),         # odd x odd, tiles straddling samples (117 pixels), a partial channel tile (160 = 128 + 32)
    (9, 40, 192, 5, 6),          # ragged Cin (padded to 64), more samples than statistic slots of a tile
    (2, 128, 256, 1, 8),         # one row
]
MODES = ["online_u8_bn_res_relu", "online_s8_bn_res_none", "offline_u8_bn_res_relu", "online_u8_bias_nores_relu6", "online_u8_w4_layer"]


def _make(case, mode, seed=0):
    n, cin, cout, h, w = case
    rng = np.random.default_rng(seed + 13 * cin + h)
    signed = "s8" in mode
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32) * np.float32(1.3)
    if not signed:
        x = np.maximum(x, 0)
    x[0, 0, 0, 0] = np.float32(4.75)
    k = dict(case=case, signed=signed, x=x, offline="offline" in mode)
    k["w"] = (rng.standard_normal((cout, cin)) * 0.2).astype(np.float32)
    k["wt_width"], k["rps"] = (4, cout) if "w4" in mode else (8, 1)
    k["res"] = None if "nores" in mode else (rng.standard_normal((n, cout, h, w)) * 2).astype(np.float32)
    # the maximum of a sample sits on a pixel that is NOT stored: the statistic must still see it
    if k["res"] is not None and h > 1 and w > 1:
        k["res"][n - 1, cout - 1, 1, 1] = np.float32(97.0)
    if "bias" in mode:
        k["bn"], k["bias"] = (None, None), (rng.standard_normal(cout) * 0.1).astype(np.float32)
    else:
        k["bn"] = ((0.5 + rng.random(cout)).astype(np.float32) * np.where(rng.random(cout) < 0.1, -1, 1).astype(np.float32),
                   (rng.standard_normal(cout) * 0.3).astype(np.float32))
        k["bias"] = None
    k["act"] = "relu6" if "relu6" in mode else ("relu" if "relu" in mode else None)
    return k


def _run(k, dev, ops, sub):
    n, cin, cout, h, w = k["case"]
    x = _t(k["x"], dev)
    flags = ops.act_flags(signed=k["signed"])
    codes, scales, rowsum = ops.weight_codes(_t(k["w"], dev), k["rps"], k["wt_width"])
    cur = torch.zeros(1, device=dev)
    xstat = ops.absmax_per_sample(x)
    plan = dict(in_thr=torch.full((1,), 3.25, device=dev), in_stat=xstat) if k["offline"] else dict(in_stat=xstat)
    y, stat = ops.pwconv_i8(x, codes, scales, rowsum, _t(k["bias"], dev), width=8, flags=flags, cur_out=cur,
                            bn_scale=_t(k["bn"][0], dev), bn_shift=_t(k["bn"][1], dev), act=k["act"], residual=_t(k["res"], dev),
                            subsample=sub, **plan)
    return N(y), N(stat), N(cur)


@pytest.mark.parametrize("case", CASES, ids=["%dx%d->%d@%dx%d" % c for c in CASES])
@pytest.mark.parametrize("mode", MODES)
def test_sub2_stores_the_even_pixels_of_the_whole_launch_and_keeps_its_statistic(dev, ops, case, mode):
    k = _make(case, mode)
    n, cin, cout, h, w = case
    y, stat, cur = _run(k, dev, ops, True)
    yf, statf, curf = _run(k, dev, ops, False)
    assert y.shape == (n, cout, (h + 1) // 2, (w + 1) // 2)
    _eq(y, yf[:, :, ::2, ::2], "stored pixels")
    _eq(stat, statf, "per-sample statistic (all of y)")
    _eq(cur, curf, "current_input_max")
    if k["res"] is not None and h > 1 and w > 1 and k["act"] != "relu6":
        assert stat[n - 1] > np.abs(y[n - 1]).max()            # (the planted maximum is on an odd pixel)
    # the host twin
    from oracle import host as H
    hy, hstat = H.pwconv_i8(k["x"], k["w"].reshape(cout, cin, 1, 1), k["rps"], k["wt_width"], in_max=3.25 if k["offline"] else None,
                            in_stat=H.absmax_per_sample(k["x"]), signed=k["signed"], bias=k["bias"], bn_scale=k["bn"][0],
                            bn_shift=k["bn"][1], act=k["act"], want_stat=True, residual=k["res"], subsample=True)
    _eq(y, hy, "host twin: stored pixels")
    _eq(stat, hstat, "host twin: statistic")


def test_sub2_refuses_what_it_is_not_built_for(dev, ops):
    x = torch.zeros(2, 160, 8, 8, device=dev)                    # (padded to 192 channels: six slabs are not instantiated)
    w = torch.ones(256, 160, device=dev)
    codes, scales, rowsum = ops.weight_codes(w, 1, 8)
    assert not ops.pwconv_sub2_supported(160, 256) and not ops.pwconv_sub2_supported(64, 128) and ops.pwconv_sub2_supported(64, 256)
    with pytest.raises(ValueError):
        ops.pwconv_i8(x, codes, scales, rowsum, in_stat=ops.absmax_per_sample(x), subsample=True)
    with pytest.raises(ValueError):
        ops.pwconv_i8(torch.zeros(2, 64, 8, 8, device=dev), *ops.weight_codes(torch.ones(256, 64, device=dev), 1, 8),
                      in_stat=torch.ones(2, device=dev), subsample=True, stride=2)


def _resnet(model, quant_type, ctx):
    from test_gpu_net import _build as build
    net = build(model, 1000, ctx, quant_type=quant_type) if quant_type else build(model, 1000, ctx)
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    return net


@pytest.mark.parametrize("size", [224, 200])
def test_resnet50_with_subsampled_stage_boundaries_equals_the_same_net_without(dev, ops, size):
    """Three launches store a quarter of their output and six stride-2 convolutions read it with stride 1: no value changes
    (size 200: 50 x 50, 25 x 25 and 13 x 13 planes - odd rows and columns at two of the three boundaries)."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    rng = np.random.default_rng(5)
    X = mx.nd.array(rng.standard_normal((4, 3, size, size)).astype(np.float32), ctx=mx.gpu(0))
    outs = {}
    for on in (False, True):
        net = _resnet("resnet50_v1", "channel", mx.gpu(0))
        net(mx.nd.NDArray(X._t[:2].contiguous()))
        fuse.fuse_inference(net)
        old = fuse.SUBSAMPLE
        fuse.SUBSAMPLE = on
        seen = []
        real = ops.pwconv_i8

        def spy(*a, **k):
            if k.get("subsample"):
                seen.append(tuple(a[0].shape))
            return real(*a, **k)
        ops.pwconv_i8 = spy
        try:
            out = net(X)
            cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
            net.update_ema()
            thr = np.asarray([b.input_max.data().asscalar() for b in net.collect_quantized_blocks()], np.float32)
        finally:
            fuse.SUBSAMPLE = old
            ops.pwconv_i8 = real
        outs[on] = (N(out._t), cur, thr, seen)
    assert len(outs[False][3]) == 0 and len(outs[True][3]) == 3, outs[True][3]
    _eq(outs[True][0], outs[False][0], "logits")
    _eq(outs[True][1], outs[False][1], "current_input_max of every block")
    _eq(outs[True][2], outs[False][2], "thresholds after one naive-EMA step")


def test_a_hook_between_producer_and_readers_keeps_the_whole_trunk_and_a_stray_reader_fails_loudly(dev, ops):
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    net = _resnet("resnet50_v1", "channel", mx.gpu(0))
    X = mx.nd.array(np.random.default_rng(2).standard_normal((2, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0))
    net(X)
    fuse.fuse_inference(net)
    want = net(X).asnumpy()
    stages = [b for b in net.features._children.values() if isinstance(b, mx.gluon.nn.HybridSequential) and len(b._children) >= 3]
    assert len(stages) == 4
    shapes = []
    h = stages[1].register_forward_pre_hook(lambda blk, inp: shapes.append(tuple(inp[0].shape)))
    try:
        got = net(X).asnumpy()
    finally:
        h.detach()
    assert shapes == [(2, 256, 56, 56)]                          # the hook saw the whole tensor
    _eq(got, want, "logits with a hook on a stage")
    # a subsampled trunk handed to a block that is not one of its readers
    first = list(list(stages[1]._children.values())[0].body._children.values())[0]
    other = list(list(stages[2]._children.values())[0].body._children.values())[0]
    fake = mx.nd.NDArray(torch.zeros(2, 512, 14, 14, device=dev))
    fake._fq_stat = torch.ones(2, device=dev)
    fake._fq_sub2 = {"hw": (28, 28), "readers": (first,), "unit": None}
    with pytest.raises(RuntimeError, match="not one of its two readers"):
        other(fake)


DUAL_CASES = [(3, 64, 256, 56, 56), (3, 128, 512, 28, 28), (5, 256, 1024, 14, 14), (3, 64, 256, 9, 11), (9, 512, 2048, 7, 7)]


@pytest.mark.parametrize("case", DUAL_CASES, ids=["%dx%d->%d@%dx%d" % c for c in DUAL_CASES])
def test_dual_sub2_stores_both_outputs_subsampled(dev, ops, case):
    """fq_pwconv_i8_c16_dual_sub2: codes in, residual added; the fp32 output and its code copy hold y[:, :, ::2, ::2] of what
    fq_pwconv_i8_c16_dual stores - same statistic - and agree with the host twin."""
    from oracle import fq_oracle as O
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + 31)
    x = np.maximum(rng.standard_normal((n, cin, h, w)) * 2, 0).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    res = (rng.standard_normal((n, cout, h, w)) * 2).astype(np.float32)
    res[n - 1, 3, h - 1 - (h % 2), 1] = np.float32(88.0)            # a maximum on an odd column
    codes, scales, rowsum = ops.weight_codes(_t(wt, dev), 1, 8)
    thr, thr2 = np.float32(2.3), np.float32(3.1)
    thr_t = _t(np.float32([thr]), dev)
    stat_in = _t(O.absmax_per_sample(x), dev)
    cx = O.ste_codes(x, O.act_scale(thr, False, 8), thr, np.float32(0))
    xc = ops.Codes16(_t(O.to_c16(cx.astype(np.int64), 128), dev), x.shape, thr_t, 8, 0)
    cur_a, cur_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    kw = dict(in_thr=thr_t, width=8, flags=0, bn_scale=_t(sc, dev), bn_shift=_t(sh, dev), act="relu", in_stat=stat_in,
              residual=_t(res, dev))
    side_kw = dict(thr=_t(np.float32([thr2]), dev), width=8, flags=0)
    want, want_stat, want_side = ops.pwconv_i8(xc, codes, scales, rowsum, cur_out=cur_a, side_codes=side_kw, **kw)
    got, got_stat, side = ops.pwconv_i8(xc, codes, scales, rowsum, cur_out=cur_b, side_codes=side_kw, subsample=True, **kw)
    hs, ws = (h + 1) // 2, (w + 1) // 2
    assert tuple(got.shape) == (n, cout, hs, ws) and side.shape == (n, cout, hs, ws)
    _eq(N(got), N(want)[:, :, ::2, ::2], "fp32 output")
    _eq(N(got_stat), N(want_stat), "statistic")
    _eq(N(cur_a), N(cur_b), "current_input_max")
    full16 = N(want_side.t).reshape(n, cout // 16, h, w, 16)
    _eq(N(side.t).reshape(n, cout // 16, hs, ws, 16), full16[:, :, ::2, ::2], "code copy")
    if h > 1:
        assert N(got_stat)[n - 1] >= 80.0


def test_resnet50_offline_with_subsampled_stage_boundaries_equals_the_same_net_without(dev, ops):
    """Offline thresholds: the last unit of a stage stores trunk and code copy subsampled (fq_pwconv_i8_c16_dual_sub2), the two
    readers take the codes (or the fp32 quarter) with stride 1: logits and every block's current_input_max bit-equal."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        rng = np.random.default_rng(8)
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
            old = fuse.SUBSAMPLE
            fuse.SUBSAMPLE = on
            seen, readers = [], []
            real = ops.pwconv_i8

            def spy(*a, **k):
                if k.get("subsample"):
                    seen.append((tuple(a[0].shape), k.get("side_codes") is not None))
                elif isinstance(a[0], ops.Codes16) and a[0].shape[1] in (256, 512, 1024) and a[0].shape[2] in (28, 14, 7) \
                        and a[0].shape[1] * a[0].shape[2] == 7168:
                    readers.append(tuple(a[0].shape))                # (256 @28x28, 512 @14x14, 1024 @7x7: a stage's input)
                return real(*a, **k)
            ops.pwconv_i8 = spy
            # (the shortcut convolutions of stages 2 to 4 read the subsampled code copy as the second operand of their unit's closing
            # 1x1: fq_pwconv_i8_shortcut_c16)
            real_short = ops.pwconv_i8_shortcut

            def spy_short(*a, **k):
                if isinstance(k.get("x2"), ops.Codes16) and k["x2"].shape[1] * k["x2"].shape[2] == 7168:
                    readers.append(tuple(k["x2"].shape))
                return real_short(*a, **k)
            ops.pwconv_i8_shortcut = spy_short
            try:
                out = net(xs[2])
                cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
            finally:
                fuse.SUBSAMPLE = old
                ops.pwconv_i8 = real
                ops.pwconv_i8_shortcut = real_short
            outs[on] = (N(out._t), cur, seen, readers)
        assert len(outs[False][2]) == 0 and len(outs[True][2]) == 3, outs[True][2]
        # both readers of every boundary take the subsampled code copy with stride 1 (256 @28x28, 512 @14x14, 1024 @7x7 are the
        # shapes of no other code tensor a 1x1 of this net reads)
        assert len(outs[True][3]) == 6 and len(outs[False][3]) == 0, outs[True][3]
        assert all(dual for _, dual in outs[True][2]), outs[True][2]          # (codes in, both outputs: the dual form)
        _eq(outs[True][0], outs[False][0], "logits")
        _eq(outs[True][1], outs[False][1], "current_input_max of every block")
    finally:
        torch.backends.cudnn.deterministic = was
