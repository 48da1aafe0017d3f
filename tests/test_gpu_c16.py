"""int8 hand-over between fused convolutions under OFFLINE input quantisation (include/fakequant.h at fq_pwconv_i8_c16): C16 code
tensors instead of fp32 NCHW between a producer and its single consumer.  The codes are a function of the same fp32 values and
the same stored threshold, so everything is checked for EQUALITY:
  * a producer's C16 output == oracle codes of its own fp32 output under the consumer's threshold, statistic unchanged;
  * a consumer fed with C16 codes == the same consumer fed with the fp32 tensor and in_thr;
  * a whole net with hand-overs == the same net without them, logits bit for bit."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(gpu):
    return gpu.torch_device


@pytest.fixture(scope="module")
def ops():
    from quantization.mxnet_amd import ops as _ops
    return _ops


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# (the first five: more than 4096 tiles of 32 pixels - C16 input and ragged channel counts then take the thin instantiations of
# the streaming form, round 4, instead of one-tile workgroups of the split form; all of them are compared with the split form)
PW_CASES = [(12, 16, 96, 112, 112, 1), (11, 32, 16, 112, 112, 1), (43, 96, 24, 56, 56, 1), (43, 24, 144, 56, 56, 1),
            (43, 144, 24, 56, 56, 1), (2, 64, 64, 14, 14, 1), (3, 16, 96, 9, 11, 1), (2, 144, 24, 7, 7, 1), (2, 96, 40, 5, 6, 1), (2, 256, 512, 8, 8, 2),
            (1, 320, 1280, 7, 7, 1), (4, 960, 160, 3, 3, 1), (1, 32, 192, 28, 28, 1),
            # (MobileNetV2's expansions on 14x14 / 7x7: fp32 in, codes out)
            (3, 64, 384, 14, 14, 1), (2, 96, 576, 14, 14, 1), (3, 160, 960, 7, 7, 1), (2, 64, 320, 5, 5, 1)]


@pytest.mark.parametrize("case", PW_CASES, ids=["%dx%d->%d@%dx%d/s%d" % c for c in PW_CASES])
@pytest.mark.parametrize("mode", ["u8-out", "s8-out", "s8-out-relu", "u8-out-relu6-thr9"])
def test_pointwise_producer_writes_the_consumers_codes(dev, ops, case, mode):
    """(the last two modes: a ReLU in front of a SIGNED consumer range, and a ReLU6 whose 6 lies below the consumer's threshold -
    the epilogue folds the activation into the clip, med3(v, 0, min(6, hi)), and takes the statistic from the raw values)"""
    signed_out = mode.startswith("s8")
    n, cin, cout, h, w, stride = case
    rng = np.random.default_rng(sum(case) + 3)
    x = np.maximum(rng.standard_normal((n, cin, h, w)) * 2, 0).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.2).astype(np.float32)
    sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), 1, 8)
    thr_in = T(np.float32([2.7]), dev)
    act = {"u8-out": "relu6", "s8-out": None, "s8-out-relu": "relu", "u8-out-relu6-thr9": "relu6"}[mode]
    if mode == "u8-out-relu6-thr9":
        sc = sc * np.float32(4)                                          # (values beyond 6, so that the 6 is what clips)
    kw = dict(in_thr=thr_in, width=8, flags=0, bn_scale=T(sc, dev), bn_shift=T(sh, dev), act=act, stride=stride)
    y, stat = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, form="split", **kw)
    thr_out = np.float32(9.0 if mode == "u8-out-relu6-thr9" else 1.9)
    oflags = ops.act_flags(signed=signed_out)
    yc, stat_c = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, out_codes=dict(thr=T(np.float32([thr_out]), dev), width=8,
                                                                                flags=oflags), **kw)
    assert isinstance(yc, ops.Codes16) and yc.shape == tuple(y.shape)
    yf = y.cpu().numpy()
    want = O.ste_codes(yf, O.act_scale(thr_out, signed_out, 8), thr_out, np.float32(-thr_out) if signed_out else np.float32(0))
    assert np.array_equal(yc.t.cpu().numpy(), O.to_c16(want.astype(np.int64), 0 if signed_out else 128)), "C16 codes"
    assert torch.equal(stat, stat_c), "the statistic is that of the fp32 values"


@pytest.mark.parametrize("case", PW_CASES, ids=["%dx%d->%d@%dx%d/s%d" % c for c in PW_CASES])
@pytest.mark.parametrize("mode", ["u8_bn_relu", "s8_res"])
def test_pointwise_consumer_of_codes_equals_consumer_of_fp32(dev, ops, case, mode):
    n, cin, cout, h, w, stride = case
    rng = np.random.default_rng(sum(case) + 5)
    signed = "s8" in mode
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    if not signed:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.2).astype(np.float32)
    sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), cout, 8)
    thr = np.float32(2.3)
    thr_t = T(np.float32([thr]), dev)
    flags = ops.act_flags(signed=signed)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    kw = dict(in_thr=thr_t, width=8, flags=flags, bn_scale=T(sc, dev), bn_shift=T(sh, dev), stride=stride,
              act="relu" if "relu" in mode else None)
    if "res" in mode:
        kw["residual"] = T((rng.standard_normal((n, cout, ho, wo)) * 3).astype(np.float32), dev)
    stat_in = T(O.absmax_per_sample(x), dev)
    cur_a, cur_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    want, want_stat = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, form="split", in_stat=stat_in, cur_out=cur_a, **kw)
    cx = O.ste_codes(x, O.act_scale(thr, signed, 8), thr, np.float32(-thr) if signed else np.float32(0))
    xc = ops.Codes16(T(O.to_c16(cx.astype(np.int64), 0 if signed else 128), dev), x.shape, thr_t, 8, flags)
    got, got_stat = ops.pwconv_i8(xc, codes, scales, rowsum, in_stat=stat_in, cur_out=cur_b, **kw)
    assert torch.equal(got, want), "output of the consumer fed with codes"
    assert torch.equal(got_stat, want_stat) and torch.equal(cur_a, cur_b)
    with pytest.raises(ValueError, match="another threshold"):
        ops.pwconv_i8(xc, codes, scales, rowsum, in_thr=T(np.float32([thr]), dev), width=8, flags=flags)


BOTH_CASES = [(3, 256, 64, 56, 56, 1), (2, 512, 128, 28, 28, 1), (5, 1024, 256, 14, 14, 1), (7, 2048, 512, 7, 7, 1), (2, 256, 64, 9, 11, 1),
              (2, 512, 128, 28, 28, 2),
              # the thin streaming form's both-sides instantiation (more than 4096 pixel tiles, Cin <= 32): MobileNetV2's
              # 32 -> 16 and 16 -> 96 at 112 x 112, and a ragged 24 -> 144
              (11, 32, 16, 112, 112, 1), (11, 16, 96, 112, 112, 1), (43, 24, 144, 56, 56, 1), (11, 32, 32, 112, 112, 1)]


@pytest.mark.parametrize("case", BOTH_CASES, ids=["%dx%d->%d@%dx%d/s%d" % c for c in BOTH_CASES])
def test_pointwise_between_two_code_tensors(dev, ops, case):
    """Codes in AND codes out (round 4: the first 1x1 of a ResNet unit fed by the trunk's code copy, handing codes to the
    unit's 3x3): == the codes of what the fp32-in / fp32-out call computes, statistic and batch mean unchanged."""
    n, cin, cout, h, w, stride = case
    rng = np.random.default_rng(sum(case) + 17)
    x = np.maximum(rng.standard_normal((n, cin, h, w)) * 2, 0).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.05).astype(np.float32)
    sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), 1, 8)
    thr, thr2 = np.float32(2.3), np.float32(1.7)
    thr_t = T(np.float32([thr]), dev)
    stat_in = T(O.absmax_per_sample(x), dev)
    cur_a, cur_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    kw = dict(in_thr=thr_t, width=8, flags=0, bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu", stride=stride, in_stat=stat_in)
    want, want_stat = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, form="split", cur_out=cur_a, **kw)
    cx = O.ste_codes(x, O.act_scale(thr, False, 8), thr, np.float32(0))
    xc = ops.Codes16(T(O.to_c16(cx.astype(np.int64), 128), dev), x.shape, thr_t, 8, 0)
    yc, st = ops.pwconv_i8(xc, codes, scales, rowsum, cur_out=cur_b, out_codes=dict(thr=T(np.float32([thr2]), dev), width=8, flags=0),
                           **kw)
    wantc = O.to_c16(O.ste_codes(want.cpu().numpy(), O.act_scale(thr2, False, 8), thr2, np.float32(0)).astype(np.int64), 128)
    assert isinstance(yc, ops.Codes16) and yc.shape == tuple(want.shape)
    assert np.array_equal(yc.t.cpu().numpy(), wantc), "codes of the output"
    assert torch.equal(st, want_stat) and torch.equal(cur_a, cur_b)


DUAL_CASES = [(43, 64, 256, 56, 56), (3, 64, 256, 9, 11), (5, 128, 512, 28, 28), (7, 256, 1024, 14, 14), (9, 512, 2048, 7, 7),
              (43, 128, 512, 56, 56)]


@pytest.mark.parametrize("case", DUAL_CASES, ids=["%dx%d->%d@%dx%d" % c for c in DUAL_CASES])
def test_closing_pointwise_stores_the_trunk_twice(dev, ops, case):
    """fq_pwconv_i8_c16_dual (round 4): codes in, residual added, fp32 out - and the codes of that output under the NEXT
    consumer's threshold beside it.  y, statistic and batch mean are those of the one-output call; the side tensor holds the
    oracle's codes of y.  (The first case runs the streaming form - more than 4096 pixel tiles -, the others the split form.)"""
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + 23)
    x = np.maximum(rng.standard_normal((n, cin, h, w)) * 2, 0).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    res = (rng.standard_normal((n, cout, h, w)) * 2).astype(np.float32)
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), 1, 8)
    thr, thr2 = np.float32(2.3), np.float32(3.1)
    thr_t = T(np.float32([thr]), dev)
    stat_in = T(O.absmax_per_sample(x), dev)
    cx = O.ste_codes(x, O.act_scale(thr, False, 8), thr, np.float32(0))
    xc = ops.Codes16(T(O.to_c16(cx.astype(np.int64), 128), dev), x.shape, thr_t, 8, 0)
    cur_a, cur_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    kw = dict(in_thr=thr_t, width=8, flags=0, bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu", in_stat=stat_in,
              residual=T(res, dev))
    want, want_stat = ops.pwconv_i8(xc, codes, scales, rowsum, cur_out=cur_a, **kw)
    got, got_stat, side = ops.pwconv_i8(xc, codes, scales, rowsum, cur_out=cur_b,
                                        side_codes=dict(thr=T(np.float32([thr2]), dev), width=8, flags=0), **kw)
    assert torch.equal(got, want) and torch.equal(got_stat, want_stat) and torch.equal(cur_a, cur_b)
    wantc = O.to_c16(O.ste_codes(want.cpu().numpy(), O.act_scale(thr2, False, 8), thr2, np.float32(0)).astype(np.int64), 128)
    assert isinstance(side, ops.Codes16) and side.shape == tuple(want.shape)
    assert np.array_equal(side.t.cpu().numpy(), wantc), "the side tensor's codes"
    # ... and a consumer fed with the side tensor computes what it computes from the fp32 trunk
    w2 = (rng.standard_normal((cout // 4, cout, 1, 1)) * 0.05).astype(np.float32)
    c2, s2, r2 = ops.weight_codes(T(w2, dev), 1, 8)
    thr2_t = side.thr
    a, _ = ops.pwconv_i8(want, c2, s2, r2, in_thr=thr2_t, width=8, flags=0, act="relu", form="split")
    b, _ = ops.pwconv_i8(side, c2, s2, r2, in_thr=thr2_t, width=8, flags=0, act="relu")
    assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(43, 224, 224), (3, 64, 64), (2, 33, 47)], ids=["43x224", "3x64", "2x33x47"])
@pytest.mark.parametrize("act", ["relu6", "relu", "relu6-thr9"])
def test_first_convolution_hands_its_consumers_codes_over(dev, ops, shape, act):
    """fq_stem_conv3x3s2_c16 (round 4): the codes of what fq_stem_conv3x3s2 computes, under the consumer's threshold; same
    statistic.  And the consumer - a 32 -> 32 1x1 that reads codes AND writes codes (the streaming form's K = 32 both-sides
    instantiation on the large plane, the split form's fp32-out path otherwise) - computes what it computes from the fp32 tensor."""
    n, h, w = shape
    rng = np.random.default_rng(sum(shape) + 29)
    x = rng.standard_normal((n, 3, h, w)).astype(np.float32)
    wt = (rng.standard_normal((32, 3, 3, 3)) * 0.3).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    sh = rng.standard_normal(32).astype(np.float32)
    thr = np.float32(2.1)
    if act == "relu6-thr9":                      # (a ReLU6 whose 6 lies below the consumer's threshold: the 6 is what clips)
        act, thr, sc = "relu6", np.float32(9.0), sc * np.float32(6)
    thr_t = T(np.float32([thr]), dev)
    y, stat = ops.stem_conv_s2(T(x, dev), T(wt, dev), None, bn_scale=T(sc, dev), bn_shift=T(sh, dev), act=act)
    yc, stat_c = ops.stem_conv_s2(T(x, dev), T(wt, dev), None, bn_scale=T(sc, dev), bn_shift=T(sh, dev), act=act,
                                  out_codes=dict(thr=thr_t, width=8, flags=0))
    assert isinstance(yc, ops.Codes16) and yc.shape == tuple(y.shape) and torch.equal(stat, stat_c)
    want = O.to_c16(O.ste_codes(y.cpu().numpy(), O.act_scale(thr, False, 8), thr, np.float32(0)).astype(np.int64), 128)
    assert np.array_equal(yc.t.cpu().numpy(), want), "codes of the first convolution's output"
    w2 = (rng.standard_normal((32, 32, 1, 1)) * 0.2).astype(np.float32)
    codes, scales, rowsum = ops.weight_codes(T(w2, dev), 1, 8)
    thr2 = np.float32(1.4)
    kw = dict(in_thr=thr_t, width=8, flags=0, bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu6", in_stat=stat)
    a, sa = ops.pwconv_i8(y, codes, scales, rowsum, form="split", **kw)
    if n * y.shape[2] * y.shape[3] > 32 * 4096:
        b, sb = ops.pwconv_i8(yc, codes, scales, rowsum, out_codes=dict(thr=T(np.float32([thr2]), dev), width=8, flags=0), **kw)
        wantb = O.to_c16(O.ste_codes(a.cpu().numpy(), O.act_scale(thr2, False, 8), thr2, np.float32(0)).astype(np.int64), 128)
        assert np.array_equal(b.t.cpu().numpy(), wantb) and torch.equal(sa, sb)
    else:
        b, sb = ops.pwconv_i8(yc, codes, scales, rowsum, **kw)
        assert torch.equal(a, b) and torch.equal(sa, sb)


DW_CASES = [(2, 96, 112, 112, 2), (2, 144, 56, 56, 1), (3, 24, 9, 11, 1), (2, 192, 28, 28, 2), (3, 384, 14, 14, 1),
            (2, 960, 7, 7, 1), (4, 40, 5, 6, 2), (1, 16, 70, 70, 1)]


@pytest.mark.parametrize("case", DW_CASES, ids=["%dx%d@%dx%d/s%d" % c for c in DW_CASES])
@pytest.mark.parametrize("signed", [False, True], ids=["u8", "s8"])
def test_depthwise_between_two_code_tensors(dev, ops, case, signed):
    """fq_dwconv3x3_c16: codes in (the threshold of this layer), codes out (the threshold of the next) == the oracle's codes of
    what fq_dwconv3x3 computes from the fp32 tensor under the same stored threshold; the statistic is that of the fp32
    values."""
    n, c, h, w, stride = case
    rng = np.random.default_rng(sum(case) + 13)
    x = (rng.standard_normal((n, c, h, w)) * 2).astype(np.float32)
    if not signed:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((c, 1, 3, 3)) * 0.4).astype(np.float32)
    sc = rng.uniform(0.3, 1.5, c).astype(np.float32)
    sh = rng.standard_normal(c).astype(np.float32)
    thr, thr2 = np.float32(2.3), np.float32(1.7)
    thr_t = T(np.float32([thr]), dev)
    flags = ops.act_flags(signed=signed)
    stat_in = T(O.absmax_per_sample(x), dev)
    cur_a, cur_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    kw = dict(stride=stride, in_stat=stat_in, in_thr=thr_t, width=8, flags=flags, bn_scale=T(sc, dev), bn_shift=T(sh, dev),
              act="relu6")
    want, want_stat = ops.dwconv3x3(T(x, dev), T(wt, dev), cur_out=cur_a, **kw)
    cx = O.ste_codes(x, O.act_scale(thr, signed, 8), thr, np.float32(-thr) if signed else np.float32(0))
    xc = ops.Codes16(T(O.to_c16(cx.astype(np.int64), 0 if signed else 128), dev), x.shape, thr_t, 8, flags)
    yc, st = ops.dwconv3x3_c16(xc, T(wt, dev), cur_out=cur_b, out_codes=dict(thr=T(np.float32([thr2]), dev), width=8, flags=0),
                               **kw)
    wantc = O.to_c16(O.ste_codes(want.cpu().numpy(), O.act_scale(thr2, False, 8), thr2, np.float32(0)).astype(np.int64), 128)
    assert yc.shape == tuple(want.shape)
    assert np.array_equal(yc.t.cpu().numpy(), wantc), "codes of the depthwise output"
    assert torch.equal(st, want_stat) and torch.equal(cur_a, cur_b)


DW_EDGE_CASES = [(2, 16, 1, 1, 1), (2, 20, 1, 5, 2), (1, 32, 2, 3, 1), (1, 32, 2, 2, 2), (2, 16, 3, 70, 2), (1, 48, 4, 4, 1),
                 (1, 16, 5, 5, 1), (1, 16, 6, 7, 2), (1, 16, 7, 66, 1), (1, 16, 9, 130, 2)]


@pytest.mark.parametrize("case", DW_EDGE_CASES, ids=["%dx%d@%dx%d/s%d" % c for c in DW_EDGE_CASES])
@pytest.mark.parametrize("epi", ["bn-relu", "bn-none-signed-out", "bias-bn-relu6", "plain", "bn-relu-signed-out", "bn-relu6-thr9"])
def test_depthwise_on_codes_every_epilogue_on_planes_shorter_than_the_prefetch(dev, ops, case, epi):
    """fq_dwconv3x3_c16 fetches its input rows two output rows ahead through rotating register sets and is instantiated per
    epilogue and output quantiser: planes of 1-9 rows (shorter than the rotation's period, rows fetched past the end), more
    than 64 columns (column tiles), every epilogue instantiation and the generic (signed-range) output quantiser - against
    fq_dwconv3x3 on the fp32 tensor + the oracle's codes."""
    n, c, h, w, stride = case
    rng = np.random.default_rng(sum(case) + 29 + len(epi))
    x = np.maximum((rng.standard_normal((n, c, h, w)) * 2).astype(np.float32), 0)
    wt = (rng.standard_normal((c, 1, 3, 3)) * 0.4).astype(np.float32)
    sc = rng.uniform(0.3, 1.5, c).astype(np.float32)
    sh = rng.standard_normal(c).astype(np.float32)
    bias = T(rng.standard_normal(c).astype(np.float32), dev) if epi == "bias-bn-relu6" else None
    act = {"bn-relu": "relu", "bn-none-signed-out": None, "bias-bn-relu6": "relu6", "plain": None, "bn-relu-signed-out": "relu",
           "bn-relu6-thr9": "relu6"}[epi]
    out_signed = epi in ("bn-none-signed-out", "bn-relu-signed-out")
    # (the last two: ReLU in front of a signed consumer range, and a ReLU6 whose 6 lies below the consumer's threshold - the
    # compile-time epilogues fold the activation into the clip and take the statistic from the raw sums)
    thr, thr2 = np.float32(2.3), np.float32(9.0 if epi == "bn-relu6-thr9" else 1.7)
    if epi == "bn-relu6-thr9":
        sc = sc * np.float32(3)
    thr_t = T(np.float32([thr]), dev)
    kw = dict(stride=stride, in_thr=thr_t, width=8, flags=ops.act_flags(signed=False), act=act)
    if epi != "plain":
        kw.update(bn_scale=T(sc, dev), bn_shift=T(sh, dev))
    want, want_stat = ops.dwconv3x3(T(x, dev), T(wt, dev), bias, **kw)
    cx = O.ste_codes(x, O.act_scale(thr, False, 8), thr, np.float32(0))
    xc = ops.Codes16(T(O.to_c16(cx.astype(np.int64), 128), dev), x.shape, thr_t, 8, ops.act_flags(signed=False))
    yc, st = ops.dwconv3x3_c16(xc, T(wt, dev), bias, out_codes=dict(thr=T(np.float32([thr2]), dev), width=8,
                                                                   flags=ops.act_flags(signed=out_signed)), **kw)
    lo = np.float32(-thr2) if out_signed else np.float32(0)
    wantc = O.to_c16(O.ste_codes(want.cpu().numpy(), O.act_scale(thr2, out_signed, 8), thr2, lo).astype(np.int64),
                     0 if out_signed else 128)
    assert yc.shape == tuple(want.shape)
    assert np.array_equal(yc.t.cpu().numpy(), wantc), "codes of the depthwise output"
    assert torch.equal(st, want_stat)


C3_CASES = [(2, 64, 64, 9, 11), (3, 128, 128, 7, 7), (2, 256, 256, 5, 6), (1, 512, 512, 7, 7), (2, 64, 128, 14, 14),
            (5, 64, 96, 3, 3), (2, 128, 160, 28, 28), (1, 64, 64, 56, 56)]


@pytest.mark.parametrize("case", C3_CASES, ids=["%dx%d->%d@%dx%d" % c for c in C3_CASES])
@pytest.mark.parametrize("signed", [False, True], ids=["u8", "s8"])
def test_dense3x3_with_codes_on_both_sides(dev, ops, case, signed):
    """The 3x3 convolution in the middle of a bottleneck: C16 in == fp32 in, C16 out == oracle codes of the fp32 output, and
    both at once == the codes of the all-fp32 call."""
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + 9)
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    if not signed:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    codes, scales, rowsum = ops.weight_codes_3x3(T(wt, dev), 1, 8)
    thr = np.float32(2.3)
    thr_t = T(np.float32([thr]), dev)
    flags = ops.act_flags(signed=signed)
    kw = dict(in_thr=thr_t, width=8, flags=flags, bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu")
    want, want_stat = ops.conv3x3_i8(T(x, dev), codes, scales, rowsum, **kw)
    cx = O.ste_codes(x, O.act_scale(thr, signed, 8), thr, np.float32(-thr) if signed else np.float32(0))
    xc = ops.Codes16(T(O.to_c16(cx.astype(np.int64), 0 if signed else 128), dev), x.shape, thr_t, 8, flags)
    got, got_stat = ops.conv3x3_i8(xc, codes, scales, rowsum, **kw)
    assert torch.equal(got, want) and torch.equal(got_stat, want_stat), "codes in"
    thr2 = np.float32(1.1)
    oc = dict(thr=T(np.float32([thr2]), dev), width=8, flags=0)
    wantc = O.to_c16(O.ste_codes(want.cpu().numpy(), O.act_scale(thr2, False, 8), thr2, np.float32(0)).astype(np.int64), 128)
    for src, what in ((T(x, dev), "codes out"), (xc, "codes in and out")):
        yc, st = ops.conv3x3_i8(src, codes, scales, rowsum, out_codes=oc, **kw)
        assert np.array_equal(yc.t.cpu().numpy(), wantc), what
        assert torch.equal(st, want_stat), what


def test_mobilenetv2_units_without_shortcut_hand_codes_to_the_next_block(gpu):
    """quantize/fuse.py: visit_unit_links - the 16-channel tensor between MobileNetV2's first two units (no shortcut on either
    side) crosses their containers as codes: at a batch whose 112 x 112 planes make more than 4096 pixel tiles the 16 -> 96
    convolution READS codes (and writes codes); logits and every batch statistic equal those of the
    net without the links bit for bit."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build
    was_deterministic = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        outs, stats, seen = {}, {}, {}
        for links in (True, False):
            fuse.UNIT_LINKS = links
            net = _build("mobilenetv2_1.0", 1000, gpu, quant_type="channel", wt=4)
            rng = np.random.default_rng(5)
            xs = [mx.nd.array(rng.standard_normal((11, 3, 224, 224)).astype(np.float32), ctx=gpu) for _ in range(3)]
            net.quantize_input(enable=True, online=True)
            for x in xs[:2]:
                net(x)
                net.update_ema()
            net.fix_params()
            net.quantize_input(enable=True, online=False)
            net(xs[2])
            fuse.fuse_inference(net)
            real = ops.pwconv_i8
            thin = []

            def spy(x, *a, **k):
                if isinstance(x, ops.Codes16) and x.shape[1] == 16:
                    thin.append((x.shape[1], k.get("out_codes") is not None))
                return real(x, *a, **k)
            ops.pwconv_i8 = spy
            try:
                outs[links] = net(xs[2]).asnumpy()
            finally:
                ops.pwconv_i8 = real
            stats[links] = [float(b.current_input_max) for b in net.collect_quantized_blocks()]
            seen[links] = thin
        assert seen[True] == [(16, True)] and seen[False] == [], seen
        assert np.array_equal(outs[True], outs[False]) and stats[True] == stats[False]
    finally:
        fuse.UNIT_LINKS = True
        torch.backends.cudnn.deterministic = was_deterministic


@pytest.mark.parametrize("model,kw", [("resnet50_v1", dict(quant_type="channel")), ("cifar_resnet20_v1", dict()),
                                      ("mobilenetv2_1.0", dict(quant_type="channel", wt=4))],
                         ids=["resnet50_v1", "cifar_resnet20_v1", "mobilenetv2_1.0-w4"])
def test_net_with_hand_overs_equals_net_without(gpu, model, kw):
    """Offline input quantisation, fused producers: every 1x1 -> 3x3 -> 1x1 chain of the units hands int8 codes over.  The
    logits equal those of the same net with the hand-over switched off BIT FOR BIT, every block's `current_input_max`
    included; under online quantisation nothing is handed over.  (MobileNetV2's classifier is excluded from quantisation, i.e.
    a LIBRARY convolution: MIOpen picks its solver by the workspace it can get, and some of them add in an order that
    varies from run to run - the same forward twice then differs in the last bit.  The library is asked for its
    deterministic algorithm while the two nets are compared.)"""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build
    was_deterministic = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    classes, hw, batch = (10, 32, 8) if model.startswith("cifar") else ((1000, 224, 2) if model.startswith("mobilenetv2")
                                                                       else (1000, 64, 4))
    net = _build(model, classes, gpu, **kw)
    rng = np.random.default_rng(3)
    xs = [mx.nd.array(rng.standard_normal((batch, 3, hw, hw)).astype(np.float32), ctx=gpu) for _ in range(3)]
    net.quantize_input(enable=True, online=True)
    for x in xs[:2]:                                   # naive-EMA calibration: thresholds
        net(x)
        net.update_ema()
    net.fix_params()
    net.quantize_input(enable=True, online=False)
    net(xs[2])                                         # the freezing forward
    fuse.fuse_inference(net)
    calls = {"c16_out": 0, "c16_in": 0}
    real_pw, real_c3, real_dw = ops.pwconv_i8, ops.conv3x3_i8, ops.dwconv3x3_c16

    def count(fn):
        def wrapped(x, *a, **k):
            calls["c16_in"] += isinstance(x, ops.Codes16)
            calls["c16_out"] += (k.get("out_codes") is not None) + (k.get("side_codes") is not None)
            calls["side"] = calls.get("side", 0) + (k.get("side_codes") is not None)
            return fn(x, *a, **k)
        return wrapped
    ops.pwconv_i8, ops.conv3x3_i8, ops.dwconv3x3_c16 = count(real_pw), count(real_c3), count(real_dw)
    # (round 6: the closing 1x1 of a stage's first unit computes the unit's shortcut convolution itself - fq_pwconv_i8_shortcut_c16 -
    # and reads that convolution's input, codes on stages 2 and 3, as its second operand)
    real_short = ops.pwconv_i8_shortcut

    def counted_short(x, *a, **k):
        calls["c16_in"] += isinstance(k.get("x2"), ops.Codes16)
        return count(real_short)(x, *a, **k)
    ops.pwconv_i8_shortcut = counted_short
    try:
        fuse.HANDOVER = True
        with_codes = net(xs[2]).asnumpy()
        cur_with_side = [float(b.current_input_max) for b in net.collect_quantized_blocks()]
        n_side = calls.get("side", 0)
        # round 4: the closing 1x1 of a ResNet-50 unit stores the trunk twice when the next unit of its stage opens with a 1x1
        # (2 + 3 + 5 + 2 pairs of consecutive units, and the 3 stage boundaries, whose consumer is a strided 1x1); switched off, the
        # logits and every batch statistic stay what they are
        assert n_side == (15 if model == "resnet50_v1" else 0), n_side
        fuse.SIDE_CODES = False
        calls.update(c16_out=0, c16_in=0, side=0)
        no_side = net(xs[2]).asnumpy()
        # (MobileNetV2: + the codes its first convolution hands to the first 1x1 - a producer this test does not spy on)
        assert calls["side"] == 0 and calls["c16_in"] == calls["c16_out"] + (1 if model.startswith("mobilenetv2") else 0)
        assert np.array_equal(no_side, with_codes), "logits with the trunk's code copy differ"
        cur_no_side = [float(b.current_input_max) for b in net.collect_quantized_blocks()]
        assert cur_no_side == cur_with_side, "batch statistics with the trunk's code copy differ"
        fuse.SIDE_CODES = True
        calls.update(c16_out=0, c16_in=0, side=0)
        with_codes = net(xs[2]).asnumpy()
        assert cur_no_side == [float(b.current_input_max) for b in net.collect_quantized_blocks()]
        cur_with = [float(b.current_input_max) for b in net.collect_quantized_blocks()]
        n_out, n_in = calls["c16_out"], calls["c16_in"]
        fuse.HANDOVER = False
        calls.update(c16_out=0, c16_in=0)
        without = net(xs[2]).asnumpy()
        cur_without = [float(b.current_input_max) for b in net.collect_quantized_blocks()]
        assert calls["c16_out"] == 0 and calls["c16_in"] == 0
        fuse.HANDOVER = True
        net.quantize_input(enable=True, online=True)
        calls.update(c16_out=0, c16_in=0)
        net(xs[2])
        assert calls["c16_out"] == 0 and calls["c16_in"] == 0, "online quantisation: the threshold is not known to the producer"
    finally:
        ops.pwconv_i8, ops.conv3x3_i8, ops.dwconv3x3_c16 = real_pw, real_c3, real_dw
        ops.pwconv_i8_shortcut = real_short
        fuse.HANDOVER = True
        fuse.SIDE_CODES = True
        torch.backends.cudnn.deterministic = was_deterministic
    # resnet50: 16 units x (1x1 -> 3x3 -> 1x1); mobilenetv2: 16 units with an expansion x (1x1 -> depthwise -> 1x1)
    # (through the depthwise layer of every unit: 2 hand-overs per unit)
    # (+ 3 for ResNet-50: the shortcut convolution of a stage's first unit reads the trunk's code copy too - it has calibrated to
    # the very threshold of the first 1x1 beside it, both having seen the same tensors)
    # (+ 1 for MobileNetV2: its first 1x1 reads the codes the first convolution handed over - that producer is not spied on)
    shared = 3 if model == "resnet50_v1" else (1 if model.startswith("mobilenetv2") else 0)
    assert n_in == n_out + shared and n_out >= {"resnet50_v1": 32, "mobilenetv2_1.0": 32}.get(model, 1), (n_out, n_in)
    assert np.array_equal(with_codes, without), "logits with int8 hand-overs differ from the fp32 hand-over"
    assert cur_with == cur_without


def test_hooked_blocks_are_never_handed_codes(gpu):
    """a forward (pre-)hook on a quantised block of a fused net under OFFLINE input quantisation must see the fp32
    activation, not a C16 code tensor: no hand-over past a hook.  `collect_feature_maps` - which hooks `x[0]` of every
    quantised block - then gives, on such a net, the histograms it gives with the hand-over switched off."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.quantize import fuse
    from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps
    from test_gpu_net import _build
    net = _build("resnet50_v1", 1000, gpu, quant_type="channel")
    rng = np.random.default_rng(3)
    xs = [mx.nd.array(rng.standard_normal((4, 3, 64, 64)).astype(np.float32), ctx=gpu) for _ in range(3)]
    for x in xs[:2]:
        net(x)
        net.update_ema()
    net.fix_params()
    net.quantize_input(enable=True, online=False)
    net(xs[2])
    fuse.fuse_inference(net)
    real = ops.pwconv_i8, ops.conv3x3_i8
    handed = {"n": 0}

    def count(fn):
        def wrapped(x, *a, **k):
            handed["n"] += isinstance(x, ops.Codes16) + (k.get("out_codes") is not None)
            return fn(x, *a, **k)
        return wrapped
    ops.pwconv_i8, ops.conv3x3_i8 = count(real[0]), count(real[1])
    try:
        net(xs[2])
        assert handed["n"] > 0, "the un-hooked net hands codes over"
        seen = []
        blocks = net.collect_quantized_blocks()
        hooks = [b.register_forward_pre_hook(lambda m, a: seen.append(a[0])) for b in blocks]
        handed["n"] = 0
        out_hooked = net(xs[2]).asnumpy()
        assert handed["n"] == 0, "a hooked consumer was handed codes"
        assert len(seen) == len(blocks)
        assert all(getattr(a, "_fq_c16", None) is None and a._t.dtype == torch.float32 for a in seen)
        for h in hooks:
            h.detach()
        assert np.array_equal(net(xs[2]).asnumpy(), out_hooked)          # hand-overs back; same logits
        assert handed["n"] > 0
        loader = [(x, None) for x in xs]
        fuse.HANDOVER = True
        h1, r1 = collect_feature_maps(net, 2048, loader, gpu)
        fuse.HANDOVER = False
        h0, r0 = collect_feature_maps(net, 2048, loader, gpu)
        for b in blocks:
            assert r1[b] == r0[b] and np.array_equal(h1[b], h0[b])
    finally:
        ops.pwconv_i8, ops.conv3x3_i8 = real
        fuse.HANDOVER = True


THIN_CASES = [(11, 24, 144, 112, 112), (11, 16, 24, 112, 112), (43, 144, 24, 56, 56), (43, 40, 72, 56, 56), (42, 160, 48, 57, 57)]


@pytest.mark.parametrize("case", THIN_CASES, ids=["%dx%d->%d@%dx%d" % c for c in THIN_CASES])
@pytest.mark.parametrize("mode", ["online_bn_relu6", "offline_res", "signed_bias"])
def test_thin_streaming_form_equals_the_split_form(dev, ops, case, mode):
    """fp32 in, fp32 out, channel counts that are no multiples of 16 / 32, on planes with more than 4096 tiles: the shape-based
    choice is the streaming form's PART instantiation (weights in LDS, persistent wavefronts, clamped loads of the ragged
    half-slab, dropped stores past Cout); it must equal the split form bit for bit - outputs, per-sample statistic, batch mean."""
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + len(mode))
    signed = mode == "signed_bias"
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    if not signed:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.2).astype(np.float32)
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), cout, 8)
    flags = ops.act_flags(signed=signed)
    kw = dict(width=8, flags=flags)
    if mode == "online_bn_relu6":
        kw.update(bn_scale=T(rng.uniform(0.3, 1.5, cout).astype(np.float32), dev),
                  bn_shift=T(rng.standard_normal(cout).astype(np.float32), dev), act="relu6")
    elif mode == "offline_res":
        kw.update(in_thr=T(np.float32([2.1]), dev), bn_scale=T(rng.uniform(0.3, 1.5, cout).astype(np.float32), dev),
                  bn_shift=T(rng.standard_normal(cout).astype(np.float32), dev),
                  residual=T((rng.standard_normal((n, cout, h, w)) * 3).astype(np.float32), dev))
    else:
        kw.update(bias=T(rng.standard_normal(cout).astype(np.float32), dev), act="relu")
    stat_in = T(O.absmax_per_sample(x), dev)
    cur_a, cur_b = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    bias = kw.pop("bias", None)
    want, want_stat = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, bias, form="split", in_stat=stat_in, cur_out=cur_a, **kw)
    got, got_stat = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, bias, in_stat=stat_in, cur_out=cur_b, **kw)
    assert torch.equal(got, want)
    assert torch.equal(got_stat, want_stat) and torch.equal(cur_a, cur_b)


def test_hand_over_logits_against_the_oracle_chain(gpu):
    """VERDICT r4 (housekeeping): the code-hand-over net compared with the ORACLE directly, not with the fp32-hand-over product
    path - the same net (same seed, the GPU net's calibrated thresholds copied over) run on the CPU through the numpy oracle's
    entry points (oracle.patch), offline input quantisation.  Quantised blocks are exact on both sides; what differs is the
    un-quantised first convolution (MIOpen vs numpy summation order), hence a tolerance, as in test_gpu_net.py."""
    from oracle.patch import oracle_ops
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build
    model, classes, hw, batch = "cifar_resnet20_v1", 10, 32, 8
    net = _build(model, classes, gpu)
    rng = np.random.default_rng(3)
    xs_np = [rng.standard_normal((batch, 3, hw, hw)).astype(np.float32) for _ in range(3)]
    xs = [mx.nd.array(a, ctx=gpu) for a in xs_np]
    net.quantize_input(enable=True, online=True)
    for x in xs[:2]:
        net(x)
        net.update_ema()
    net.fix_params()
    net.quantize_input(enable=True, online=False)
    net(xs[2])
    fuse.fuse_inference(net)
    handed = {"n": 0}
    real = ops.conv3x3_i8

    def spy(x, *a, **k):
        handed["n"] += isinstance(x, ops.Codes16) or k.get("out_codes") is not None
        return real(x, *a, **k)
    ops.conv3x3_i8 = spy
    try:
        got = net(xs[2]).asnumpy()
    finally:
        ops.conv3x3_i8 = real
    assert handed["n"] > 0, "no code hand-over took place: the comparison would not cover it"
    thresholds = [b.input_max.data().asnumpy().copy() for b in net.collect_quantized_blocks()]
    with oracle_ops():
        ref_net = _build(model, classes, mx.cpu())
        ref_net.quantize_input(enable=True, online=True)
        ref_net(mx.nd.array(xs_np[0]))                                   # (allocates the calibration state)
        for b, t in zip(ref_net.collect_quantized_blocks(), thresholds):
            b.input_max.set_data(mx.nd.array(t))
        ref_net.fix_params()
        ref_net.quantize_input(enable=True, online=False)
        want = ref_net(mx.nd.array(xs_np[2])).asnumpy()
    # (a code that flips in an early layer - the first convolution's last bit decides it - moves a logit by a few hundredths)
    np.testing.assert_allclose(got, want, rtol=2e-2, atol=1e-1)
    assert np.abs(got - want).mean() < 3e-2


@pytest.mark.parametrize("cin,cout,hw,n", [(256, 64, 56, 32), (512, 128, 28, 128)], ids=["256->64@56x56", "512->128@28x28"])
@pytest.mark.parametrize("act", ["relu", None])
def test_wide_codes_in_codes_out_streaming_form_equals_the_split_form(dev, ops, cin, cout, hw, n, act):
    """Round 6: the first 1x1 of a ResNet-50 unit on the 56x56 / 28x28 planes, codes in and codes out, takes the streaming form from
    3000 pixel tiles up (K = 256 / 512 instantiations).  Under a stored threshold every sample is independent: the first samples of
    the full batch (streaming form) must equal the same samples run alone (few tiles: split form) - codes, statistic."""
    rng = torch.Generator(device="cpu").manual_seed(cin + hw)
    t16 = torch.randint(-128, 128, (n, cin // 16, hw * hw, 16), generator=rng, dtype=torch.int8).to(dev)
    thr = torch.full((1,), 2.75, device=dev)
    othr = torch.full((1,), 1.9, device=dev)
    w = torch.randn(cout, cin, generator=rng).to(dev) * 0.05
    codes, scales, rowsum = ops.weight_codes(w, 1, 8)
    sc = (torch.rand(cout, generator=rng) + 0.5).to(dev)
    sh = torch.randn(cout, generator=rng).to(dev) * 0.2
    kw = dict(in_thr=thr, width=8, flags=0, bn_scale=sc, bn_shift=sh, act=act, out_codes=dict(thr=othr, width=8, flags=0))
    full, full_stat = ops.pwconv_i8(ops.Codes16(t16, (n, cin, hw, hw), thr, 8, 0), codes, scales, rowsum, **kw)
    k = 3
    part, part_stat = ops.pwconv_i8(ops.Codes16(t16[:k].contiguous(), (k, cin, hw, hw), thr, 8, 0), codes, scales, rowsum, **kw)
    assert isinstance(full, ops.Codes16) and full.shape == (n, cout, hw, hw)
    assert torch.equal(full.t[:k], part.t), "codes of the first samples"
    assert torch.equal(full_stat[:k], part_stat)
    # ... and the codes are those of the fp32 output under the consumer's threshold (oracle quantiser on the fp32 twin of the call)
    yf, _ = ops.pwconv_i8(ops.Codes16(t16[:k].contiguous(), (k, cin, hw, hw), thr, 8, 0), codes, scales, rowsum,
                          in_thr=thr, width=8, flags=0, bn_scale=sc, bn_shift=sh, act=act)
    want = O.to_c16(O.ste_codes(yf.cpu().numpy(), O.act_scale(np.float32(1.9), False, 8), np.float32(1.9), np.float32(0)).astype(np.int64), 128)
    assert np.array_equal(part.t.cpu().numpy(), want)


def test_dense3x3_on_codes_with_eight_wavefronts_equals_the_four_wavefront_form(dev, ops):
    """Round 6: 256 -> 256 @14x14, codes in and codes out, takes eight wavefronts per workgroup once the layer fills the chip (batch
    96: 294 blocks of 64 pixels).  Under a stored threshold samples are independent and a block of 64 flattened pixels never
    reaches more than one sample back: the first samples of the full batch equal the same samples run alone (four wavefronts)."""
    n, c, hw = 96, 256, 14
    rng = torch.Generator(device="cpu").manual_seed(99)
    t16 = torch.randint(-128, 128, (n, c // 16, hw * hw, 16), generator=rng, dtype=torch.int8).to(dev)
    thr = torch.full((1,), 3.1, device=dev)
    othr = torch.full((1,), 2.2, device=dev)
    w = torch.randn(c, c, 3, 3, generator=rng).to(dev) * 0.02
    codes, scales, rowsum = ops.weight_codes_3x3(w, 1, 8)
    sc = (torch.rand(c, generator=rng) + 0.5).to(dev)
    sh = torch.randn(c, generator=rng).to(dev) * 0.2
    kw = dict(in_thr=thr, width=8, flags=0, bn_scale=sc, bn_shift=sh, act="relu", out_codes=dict(thr=othr, width=8, flags=0))
    full, full_stat = ops.conv3x3_i8(ops.Codes16(t16, (n, c, hw, hw), thr, 8, 0), codes, scales, rowsum, **kw)
    k = 5
    part, part_stat = ops.conv3x3_i8(ops.Codes16(t16[:k].contiguous(), (k, c, hw, hw), thr, 8, 0), codes, scales, rowsum, **kw)
    assert torch.equal(full.t[:k], part.t) and torch.equal(full_stat[:k], part_stat)
    # ... and the last samples against the same samples alone (the tail of the flattened pixel order)
    tail, tail_stat = ops.conv3x3_i8(ops.Codes16(t16[-k:].contiguous(), (k, c, hw, hw), thr, 8, 0), codes, scales, rowsum, **kw)
    assert torch.equal(full.t[-k:], tail.t) and torch.equal(full_stat[-k:], tail_stat)
