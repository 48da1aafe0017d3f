"""nn.Conv2D(quantized=True) through the C ABI (fq_qconv2d_forward) on the MI355X against the oracle - bit for bit.

The reference block (nn/quantized_conv.py:106-159) quantises input and weight per TENSOR (global range, no zero point, no
epsilon), correlates the integer codes, adds int32 bias codes, applies the activation to the integers and dequantises.  The
library does that in one call with the quantiser on the convolution's loads (matrix-core kernels for 1x1 and dense 3x3,
the depthwise forms for depthwise 3x3, an exact direct kernel for everything else).  Oracles: `oracle.fq_oracle.qconv2d_forward`
(numpy, small cases, pinned by golden G8 = the reference's own block) and its C++ twin `oracle.host.qconv2d_forward`
(pinned to the numpy oracle in tests/test_host_oracle.py) for the layer shapes of the reference's quantized MobileNet.
"""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O
from oracle import host as H

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


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def run(ops, dev, x, w, b, stride, pad, groups, **kw):
    wt = T(w, dev)
    wdt = kw.pop("weight_dtype", "int8")
    wr = kw.pop("weight_range", None)
    wbuf = ops.qconv_weights(wt, stride, pad, groups, wdt, wr)
    ws = ops.qconv_workspace(w.shape[0], dev)
    direct = kw.pop("force_direct", False) or wdt != "int8" or wr is not None
    outs = []
    for _ in range(2):                       # twice: the workspace must come back initialised
        outs.append(ops.qconv2d(T(x, dev), wt, wbuf, None if b is None else T(b, dev), stride, pad, groups, ws,
                                force_direct=direct, **kw))
    torch.cuda.synchronize()
    a, c = outs
    if isinstance(a, tuple):
        assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
        return a[0].cpu().numpy(), a[1].cpu().numpy()
    assert torch.equal(a, c)
    return a.cpu().numpy()


def relu_like(rng, shape, scale=2.0):
    return (np.maximum(rng.standard_normal(shape), 0) * scale).astype(np.float32)


# ---- the reference's own block (golden G8) ------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_bias", [0, 1])
@pytest.mark.parametrize("groups", [1, 2])
def test_golden_g8_through_the_one_call_entry_point(dev, ops, golden, use_bias, groups):
    g = golden("g8_quantized_conv")
    tag = "conv_b%d_g%d" % (use_bias, groups)
    x, w = g[tag + "/x"], g[tag + "/w"]
    b = g[tag + "/b"] if use_bias else None
    assert ops.qconv_kind(w.shape, (1, 1), (1, 1), groups) == "direct"
    y = run(ops, dev, x, w, b, (1, 1), (1, 1), groups, input_dtype="uint8")
    np.testing.assert_array_equal(y, g[tag + "/y_int"])


def test_golden_g8_through_the_block(gpu, golden):
    from quantization.mxnet_amd import mx, nn as qnn
    g = golden("g8_quantized_conv")
    for use_bias in (0, 1):
        for groups in (1, 2):
            tag = "conv_b%d_g%d" % (use_bias, groups)
            c = qnn.Conv2D(10, 3, 1, 1, in_channels=2, groups=groups, use_bias=bool(use_bias), quantized=True,
                           input_dtype="uint8", weight_dtype="int8")
            c.initialize(ctx=gpu)
            c.weight.set_data(mx.nd.array(g[tag + "/w"], ctx=gpu))
            if use_bias:
                c.bias.set_data(mx.nd.array(g[tag + "/b"], ctx=gpu))
            y = c(mx.nd.array(g[tag + "/x"], ctx=gpu)).asnumpy()
            np.testing.assert_array_equal(y, g[tag + "/y_int"])


# ---- every geometry of the block on the direct kernel, against the numpy oracle ---------------------------------------------
DIRECT_CASES = [
    # n, cin, h, w, cout, k, stride, pad, groups
    (2, 4, 9, 7, 6, (3, 3), (1, 1), (0, 0), 1),
    (2, 4, 9, 7, 6, (3, 3), (2, 2), (1, 1), 2),
    (1, 6, 11, 11, 9, (5, 5), (2, 1), (2, 2), 3),
    (3, 3, 8, 8, 5, (1, 1), (2, 2), (0, 0), 1),
    (2, 8, 6, 10, 8, (3, 1), (1, 1), (1, 0), 8),
    (1, 5, 12, 12, 7, (7, 7), (2, 2), (3, 3), 1),
]


@pytest.mark.parametrize("case", DIRECT_CASES, ids=[str(c) for c in DIRECT_CASES])
@pytest.mark.parametrize("in_dt,w_dt", [("uint8", "int8"), ("int8", "int8"), ("uint8", "uint8"), ("int8", "uint8")])
def test_direct_kernel_every_geometry(dev, ops, case, in_dt, w_dt):
    n, cin, h, w_, cout, k, st, pad, groups = case
    rng = np.random.default_rng(n * 1000 + cin * 100 + h + len(in_dt) * 7 + len(w_dt) * 13 + k[0])
    x = (rng.standard_normal((n, cin, h, w_)) * 1.5).astype(np.float32)
    if in_dt == "uint8":
        x = x + np.float32(0.4)                        # negative AND positive values: codes from L < 0 up
    w = (rng.standard_normal((cout, cin // groups) + k) * 0.2).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.5).astype(np.float32)
    for bias in (None, b):
        for act in ("none", "relu"):
            want = O.qconv2d_forward(x, w, bias, st, pad, groups, input_dtype=in_dt, weight_dtype=w_dt,
                                     act=None if act == "none" else act)
            got = run(ops, dev, x, w, bias, st, pad, groups, input_dtype=in_dt, weight_dtype=w_dt, act=act)
            np.testing.assert_array_equal(got, want)
            twin = H.qconv2d_forward(x, w, bias, st, pad, groups, input_dtype=in_dt, weight_dtype=w_dt,
                                     act=None if act == "none" else act)
            np.testing.assert_array_equal(twin, want)


def test_fixed_ranges(dev, ops):
    """`_input_range` / `_weight_range` (nn/quantized_conv.py:112-120): clip to the given range instead of the tensor's."""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((2, 4, 8, 8)) * 2).astype(np.float32)
    w = (rng.standard_normal((6, 4, 3, 3)) * 0.3).astype(np.float32)
    for ir, wr in (((-1.5, 1.5), None), ((0.0, 2.0), None), ((-0.5, 2.5), (-0.4, 0.4)), ((1.0, 2.0), None)):
        want = O.qconv2d_forward(x, w, None, (1, 1), (1, 1), 1, input_range=ir, weight_range=wr)
        got = run(ops, dev, x, w, None, (1, 1), (1, 1), 1, input_range=ir, weight_range=wr)
        np.testing.assert_array_equal(got, want)


# ---- the fused kernels: pointwise, dense 3x3, depthwise ------------------------------------------------------------------------
def fused_case(ops, dev, n, cin, h, w_, cout, k, st, pad, groups, kind, rng, in_dt="uint8", x=None, bias=False, act="none",
               wscale=0.1, **kw):
    if x is None:
        x = relu_like(rng, (n, cin, h, w_)) if in_dt == "uint8" else (rng.standard_normal((n, cin, h, w_)) * 2).astype(np.float32)
    w = (rng.standard_normal((cout, cin // groups) + k) * wscale).astype(np.float32)
    b = (rng.standard_normal(cout) * 3).astype(np.float32) if bias else None
    assert ops.qconv_kind(w.shape, st, pad, groups) == kind
    want = H.qconv2d_forward(x, w, b, st, pad, groups, input_dtype=in_dt, act=None if act == "none" else act, **kw)
    got = run(ops, dev, x, w, b, st, pad, groups, input_dtype=in_dt, act=act, **kw)
    np.testing.assert_array_equal(got, want)
    return x, w, b, want


PW_CASES = [
    # (n, cin, hw, cout): shapes of every pointwise form (stream: large planes; sample: 14x14 / 28x28; split: the rest)
    (4, 32, 112, 64), (3, 64, 56, 128), (2, 128, 56, 128), (5, 128, 28, 256), (3, 256, 28, 256), (6, 256, 14, 512),
    (9, 512, 14, 512), (20, 512, 7, 1024), (20, 1024, 7, 1024), (3, 24, 20, 40), (2, 96, 9, 200), (1, 320, 7, 1280),
]


@pytest.mark.parametrize("case", PW_CASES, ids=[str(c) for c in PW_CASES])
def test_pointwise_on_the_matrix_cores(dev, ops, case):
    n, cin, hw, cout = case
    rng = np.random.default_rng(cin * 131 + hw)
    fused_case(ops, dev, n, cin, hw, hw, cout, (1, 1), (1, 1), (0, 0), 1, "pointwise", rng)
    fused_case(ops, dev, n, cin, hw, hw, cout, (1, 1), (1, 1), (0, 0), 1, "pointwise", rng, bias=True, act="relu")
    fused_case(ops, dev, n, cin, hw, hw, cout, (1, 1), (1, 1), (0, 0), 1, "pointwise", rng, in_dt="int8", bias=True)


def test_pointwise_with_codes_away_from_zero(dev, ops):
    """uint8 without padding: the range is [min x, max x] wherever that is - negative minimum (codes from L < 0), positive
    minimum (codes from L > 0) - and the +-L re-centring of the stored bytes must cancel exactly."""
    rng = np.random.default_rng(17)
    for shift in (-0.7, 0.9, 3.0, 40.0):
        x = (rng.random((3, 64, 14, 14)) * 2 + shift).astype(np.float32)
        fused_case(ops, dev, 3, 64, 14, 14, 96, (1, 1), (1, 1), (0, 0), 1, "pointwise", rng, x=x, bias=True)


def test_accumulators_beyond_2_to_24(dev, ops):
    """1024 channels of large codes: |sum| passes 2^24, where the reference's fp32 dot stops being exact (:140-144)."""
    rng = np.random.default_rng(3)
    x = (rng.random((2, 1024, 7, 7)) * 0.2 + 0.8).astype(np.float32)
    x[0, 0, 0, 0] = 0.0
    w = (rng.random((64, 1024, 1, 1)) * 0.1 + 0.9).astype(np.float32)
    want = H.qconv2d_forward(x, w, None, (1, 1), (0, 0), 1)
    got = run(ops, dev, x, w, None, (1, 1), (0, 0), 1)
    np.testing.assert_array_equal(got, want)
    xi, xs = O.quantize_codes(x, "uint8")
    wi, wsc = O.quantize_codes(w, "int8")
    assert np.abs(np.einsum("nchw,oc->nohw", xi.astype(np.int64), wi[:, :, 0, 0].astype(np.int64))).max() > 2 ** 24


C3_CASES = [(2, 64, 14, 14, 64), (3, 64, 9, 13, 96), (1, 128, 28, 28, 128), (5, 256, 7, 7, 256), (2, 512, 7, 7, 512),
            (130, 64, 7, 7, 32)]


@pytest.mark.parametrize("case", C3_CASES, ids=[str(c) for c in C3_CASES])
def test_dense3x3_on_the_matrix_cores(dev, ops, case):
    n, cin, h, w_, cout = case
    rng = np.random.default_rng(cin + h)
    fused_case(ops, dev, n, cin, h, w_, cout, (3, 3), (1, 1), (1, 1), 1, "dense3x3", rng, wscale=0.05)
    fused_case(ops, dev, n, cin, h, w_, cout, (3, 3), (1, 1), (1, 1), 1, "dense3x3", rng, wscale=0.05, bias=True, act="relu")
    x = (rng.standard_normal((n, cin, h, w_)) * 2 + 0.5).astype(np.float32)        # uint8 range with a negative minimum
    fused_case(ops, dev, n, cin, h, w_, cout, (3, 3), (1, 1), (1, 1), 1, "dense3x3", rng, x=x, wscale=0.05)
    fused_case(ops, dev, n, cin, h, w_, cout, (3, 3), (1, 1), (1, 1), 1, "dense3x3", rng, in_dt="int8", wscale=0.05)


DW_CASES = [(2, 32, 112, 112, 1), (2, 64, 112, 112, 2), (3, 128, 56, 56, 1), (3, 128, 56, 56, 2), (4, 256, 28, 28, 1),
            (4, 256, 28, 28, 2), (9, 512, 14, 14, 1), (9, 512, 14, 14, 2), (16, 1024, 7, 7, 1), (2, 24, 19, 23, 1),
            (2, 24, 19, 23, 2), (3, 40, 10, 10, 1)]


@pytest.mark.parametrize("case", DW_CASES, ids=[str(c) for c in DW_CASES])
def test_depthwise_on_integer_codes(dev, ops, case):
    n, c, h, w_, s = case
    rng = np.random.default_rng(c + h + s)
    fused_case(ops, dev, n, c, h, w_, c, (3, 3), (s, s), (1, 1), c, "depthwise3x3", rng, wscale=0.5)
    fused_case(ops, dev, n, c, h, w_, c, (3, 3), (s, s), (1, 1), c, "depthwise3x3", rng, wscale=0.5, act="relu")
    fused_case(ops, dev, n, c, h, w_, c, (3, 3), (s, s), (1, 1), c, "depthwise3x3", rng, wscale=0.5, in_dt="int8")
    x = (rng.standard_normal((n, c, h, w_)) * 2 - 0.3).astype(np.float32)
    fused_case(ops, dev, n, c, h, w_, c, (3, 3), (s, s), (1, 1), c, "depthwise3x3", rng, x=x, wscale=0.5)
    # a bias on a depthwise layer: int32 codes of any size -> the direct kernel
    fused_case(ops, dev, n, c, h, w_, c, (3, 3), (s, s), (1, 1), c, "depthwise3x3", rng, wscale=0.5, bias=True)


# ---- where the 8-bit representation of the fast kernels does not hold: the conditional exact recomputation ------------------
def test_code_span_of_257_values_is_recomputed_exactly(dev, ops):
    """min = -0.5, max = 254.5: scale 1, L = round(-0.5) = -1, H = round(254.5) = 255 (half away from zero on both ends):
    257 codes do not fit a byte.  The record flags it and the exact direct kernel rewrites the layer."""
    rng = np.random.default_rng(1)
    x = (rng.random((2, 64, 14, 14)) * 255 - 0.5).astype(np.float32)
    x[0, 0, 0, 0], x[0, 0, 0, 1] = -0.5, 254.5
    xi, sc = O.quantize_codes(x, "uint8")
    assert sc == np.float32(1.0) and xi.min() == -1 and xi.max() == 255
    fused_case(ops, dev, 2, 64, 14, 14, 64, (1, 1), (1, 1), (0, 0), 1, "pointwise", rng, x=x, bias=True)


def test_fixed_range_that_excludes_the_padding_zero(dev, ops):
    """`_input_range` = (1, 2) on a padded 3x3: the padding zeros are clipped to 1, i.e. to a non-zero code (:108-116); the
    fused kernels assume code(0) = 0, the record flags the layer and the direct kernel rewrites it."""
    rng = np.random.default_rng(2)
    x = (rng.random((2, 16, 14, 14)) * 1.5 + 0.8).astype(np.float32)
    fused_case(ops, dev, 2, 16, 14, 14, 16, (3, 3), (1, 1), (1, 1), 16, "depthwise3x3", rng, x=x, wscale=0.5,
               input_range=(1.0, 2.0))
    x = (rng.random((2, 64, 7, 7)) * 1.5 + 0.8).astype(np.float32)
    fused_case(ops, dev, 2, 64, 7, 7, 32, (3, 3), (1, 1), (1, 1), 1, "dense3x3", rng, x=x, wscale=0.05, input_range=(1.0, 2.0))


def test_range_far_from_zero_on_the_depthwise_form(dev, ops):
    """codes around 25 000 (`_input_range` = (100, 101)): 9 * 127 * |code| passes 2^24, the fp32 chain of the depthwise form
    is no longer exact (and the padding zero is clipped to 100) -> flagged, recomputed."""
    rng = np.random.default_rng(4)
    x = (rng.random((2, 32, 14, 14)) + 100).astype(np.float32)
    fused_case(ops, dev, 2, 32, 14, 14, 32, (3, 3), (1, 1), (1, 1), 32, "depthwise3x3", rng, x=x, wscale=0.5,
               input_range=(100.0, 101.0))


# ---- project additions: the producer's statistic as the range, a BatchNorm folded behind the block --------------------------
def test_range_from_the_producers_statistic_and_folded_batchnorm(dev, ops):
    rng = np.random.default_rng(6)
    n, cin, hw, cout = 4, 128, 14, 256
    x = relu_like(rng, (n, cin, hw, hw))
    stat = np.abs(x).reshape(n, -1).max(axis=1).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 1, 1)) * 0.1).astype(np.float32)
    bsc = (rng.random(cout) + 0.5).astype(np.float32)
    bsh = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    # int8 mode: [-max, max] is what the statistic gives, padding or not
    want, wstat = H.qconv2d_forward(x, w, None, (1, 1), (0, 0), 1, input_dtype="int8", act="relu", bn_scale=bsc, bn_shift=bsh,
                                    in_stat=stat, want_stat=True)
    ref = H.qconv2d_forward(x, w, None, (1, 1), (0, 0), 1, input_dtype="int8", act="relu", bn_scale=bsc, bn_shift=bsh)
    np.testing.assert_array_equal(want, ref)
    got, gstat = run(ops, dev, x, w, None, (1, 1), (0, 0), 1, input_dtype="int8", act="relu", bn_scale=T(bsc, dev),
                     bn_shift=T(bsh, dev), in_stat=T(stat, dev), want_stat=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(gstat, wstat)
    # a padded depthwise consumer of a non-negative tensor: [0, max] from the statistic == the range pass
    wd = (rng.standard_normal((cin, 1, 3, 3)) * 0.5).astype(np.float32)
    want = H.qconv2d_forward(x, wd, None, (2, 2), (1, 1), cin)
    got = run(ops, dev, x, wd, None, (2, 2), (1, 1), cin, in_stat=T(stat, dev))
    np.testing.assert_array_equal(got, want)


def test_bad_arguments_fail_loudly(dev, ops):
    x = torch.zeros(1, 4, 5, 5, device=dev)
    w = torch.zeros(4, 4, 3, 3, device=dev)
    wbuf = ops.qconv_weights(w, (1, 1), (1, 1), 1)
    ws = ops.qconv_workspace(4, dev)
    with pytest.raises(ValueError, match="unknown out type"):
        ops.qconv2d(x, w, wbuf, None, (1, 1), (1, 1), 1, ws, input_dtype="int4")
    with pytest.raises(ValueError, match="unknown out type"):
        ops.qconv_weights(w, (1, 1), (1, 1), 1, "fp8")
    with pytest.raises(ValueError, match="channels"):
        ops.qconv2d(torch.zeros(1, 3, 5, 5, device=dev), w, wbuf, None, (1, 1), (1, 1), 1, ws)
    from quantization.mxnet_amd._lib import FakeQuantError
    with pytest.raises(FakeQuantError):
        ops.qconv2d(x.cpu(), w, wbuf, None, (1, 1), (1, 1), 1, ws)


# ---- records that carry the fix-up flag behind the fast kernels ----------------------------------------------------
@pytest.mark.parametrize("geom", ["pw", "c3", "dw"])
def test_all_zero_input_fast_kernels_equal_the_exact_kernel(dev, ops, geom):
    """A dead-ReLU tensor gives scale 0 (the record is flagged non-finite).  With an int8 input, or the producer's statistic
    on a padded layer, the conditional fix-up launch is skipped - the fast kernels' values must then BE the exact kernel's
    (every code is (int)NaN = 0 on both sides, the dequantisation factor is 0)."""
    rng = np.random.default_rng(11)
    n, c = 3, 64
    if geom == "pw":
        w, stride, pad, groups = rng.standard_normal((128, c, 1, 1)).astype(np.float32), (1, 1), (0, 0), 1
    elif geom == "c3":
        w, stride, pad, groups = rng.standard_normal((64, c, 3, 3)).astype(np.float32), (1, 1), (1, 1), 1
    else:
        w, stride, pad, groups = rng.standard_normal((c, 1, 3, 3)).astype(np.float32), (1, 1), (1, 1), c
    x = np.zeros((n, c, 14, 14), np.float32)
    bsc = rng.uniform(0.5, 1.5, w.shape[0]).astype(np.float32)
    bsh = rng.standard_normal(w.shape[0]).astype(np.float32)
    for kw in (dict(input_dtype="int8"), dict(input_dtype="uint8", in_stat=T(np.zeros(n, np.float32), dev))):
        if geom == "pw" and "in_stat" in kw:
            continue                                                   # (no padding: that call keeps its conditional launch)
        common = dict(act="relu", bn_scale=T(bsc, dev), bn_shift=T(bsh, dev), want_stat=True, **kw)
        fast, fstat = run(ops, dev, x, w, None, stride, pad, groups, **common)
        exact = run(ops, dev, x, w, None, stride, pad, groups, force_direct=True, **common)
        exact = exact[0] if isinstance(exact, tuple) else exact
        np.testing.assert_array_equal(fast, exact)
        np.testing.assert_array_equal(fast, np.broadcast_to(np.maximum(bsh, 0).reshape(1, -1, 1, 1), fast.shape))
        np.testing.assert_array_equal(fstat, np.full(n, np.maximum(bsh, 0).max(), np.float32))


def test_asymmetric_weight_codes_take_the_exact_kernel_whatever_the_caller_asks(dev, ops):
    """uint8 weights are not on one symmetric int8 grid.  The library remembers how a weight buffer was prepared: a C caller
    that passes force_direct = 0 for such a buffer still gets the exact kernel (the Python block always asked for it)."""
    rng = np.random.default_rng(12)
    x = relu_like(rng, (2, 64, 14, 14))
    w = rng.uniform(0.0, 1.0, (128, 64, 1, 1)).astype(np.float32)
    wt = T(w, dev)
    wbuf = ops.qconv_weights(wt, (1, 1), (0, 0), 1, "uint8", None)
    ws = ops.qconv_workspace(128, dev)
    got = ops.qconv2d(T(x, dev), wt, wbuf, None, (1, 1), (0, 0), 1, ws, input_dtype="uint8", force_direct=False)
    want = H.qconv2d_forward(x, w, None, (1, 1), (0, 0), 1, input_dtype="uint8", weight_dtype="uint8")
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_unaligned_1x1_input_takes_the_exact_kernel(dev, ops):
    """a contiguous tensor that does not start on a 16-byte boundary (a slice of an odd-sized buffer) is a legal
    input of the block; the one-launch pointwise forms want aligned rows, so the layer must fall back to the exact direct
    kernel instead of failing."""
    rng = np.random.default_rng(13)
    n, c, h, w_ = 2, 64, 7, 7
    x = relu_like(rng, (n, c, h, w_))
    w = rng.standard_normal((128, c, 1, 1)).astype(np.float32)
    big = torch.empty(1 + x.size, device=dev)
    xt = big[1:].view(n, c, h, w_)
    xt.copy_(T(x, dev))
    assert xt.data_ptr() % 16 == 4 and xt.is_contiguous()
    wt = T(w, dev)
    wbuf = ops.qconv_weights(wt, (1, 1), (0, 0), 1)
    ws = ops.qconv_workspace(128, dev)
    got = ops.qconv2d(xt, wt, wbuf, None, (1, 1), (0, 0), 1, ws, input_dtype="uint8")
    want = H.qconv2d_forward(x, w, None, (1, 1), (0, 0), 1, input_dtype="uint8")
    np.testing.assert_array_equal(got.cpu().numpy(), want)
