"""GPU parity: the HIP path (through the C ABI, via quantization.mxnet_amd.ops) against
  (a) the committed golden vectors made by the reference's own Python, and
  (b) the CPU oracle on seeded inputs, and
  (c) size-independent properties at BASELINE.json's full tensor sizes.
Bar: integer stage bit-exact; dequantised fp32 bit-exact (the op order is fully defined), which is tighter than the
1e-6 relative the north star allows — the looser bound is used only where a different but valid summation order exists
(none on this path today)."""
import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6      # north-star tolerance on the dequantised float


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from quantization.mxnet_amd import ops
    info = ops.device_info()
    assert info["wavefront"] == 64, info
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def ops():
    from quantization.mxnet_amd import ops as _ops
    return _ops


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def _eq(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    assert not bad.any(), "%s: %d/%d mismatches; first at %s: got %r want %r" % (
        what, int(bad.sum()), a.size, np.argwhere(bad)[0], a[bad][0], b[bad][0])


def _flags(ops, tag_signed, dense=False):
    return ops.act_flags(signed=tag_signed, lo_neg_max=False if dense else None)


# ---- activations vs golden (G4) ------------------------------------------------------------------------------
def test_conv_activation_golden(golden, dev, ops):
    g = golden("g4_activation")
    tags = sorted({k.split("/")[0] for k in g if k.startswith("conv_") and not k.startswith("conv_zero")})
    assert len(tags) == 12
    for tag in tags:
        signed = "_s_" in tag
        width = int(tag.rsplit("w", 1)[1])
        x = T(g[tag + "/x"], dev)
        y, cur, codes = ops.fake_quant_online(x, width, _flags(ops, signed), want_codes=True)
        assert N(cur)[0] == g[tag + "/online_max"], tag
        _eq(N(codes), g[tag + "/online_codes"].astype(np.int32), tag + " online codes")
        _eq(N(y), g[tag + "/online_y"], tag + " online y")
        thr = T(np.float32([g[tag + "/offline_thr"]]), dev)
        y, cur, codes = ops.fake_quant_offline(x, thr, width, _flags(ops, signed), want_codes=True)
        assert N(cur)[0] == g[tag + "/offline_curmax"], tag
        _eq(N(codes), g[tag + "/offline_codes"].astype(np.int32), tag + " offline codes")
        _eq(N(y), g[tag + "/offline_y"], tag + " offline y")
        y2, cur2, _ = ops.fake_quant_offline(x, thr, width, _flags(ops, signed), want_stat=False)
        assert cur2 is None
        _eq(N(y2), g[tag + "/offline_y"], tag + " offline y (no stat)")


def test_conv_activation_all_zero(golden, dev, ops):
    g = golden("g4_activation")
    y, cur, _ = ops.fake_quant_online(T(g["conv_zero/x"], dev))
    assert N(cur)[0] == 0.0
    _eq(N(y), g["conv_zero/online_y"])


def test_dense_activation_golden(golden, dev, ops):
    g = golden("g4_activation")
    for tag in sorted({k.split("/")[0] for k in g if k.startswith("dense_")}):
        signed = "_s_" in tag
        width = int(tag.rsplit("w", 1)[1])
        x = T(g[tag + "/x"], dev)
        y, cur, _ = ops.fake_quant_online(x, width, _flags(ops, signed, dense=True))
        assert N(cur)[0] == g[tag + "/online_max"]
        _eq(N(y), g[tag + "/online_y"], tag)
        y, _, _ = ops.fake_quant_offline(x, T(np.float32([g[tag + "/offline_thr"]]), dev), width,
                                         _flags(ops, signed, dense=True))
        _eq(N(y), g[tag + "/offline_y"], tag + " offline")


# ---- activations vs oracle on seeded shapes (ragged, unaligned, tiny, multi-chunk) ----------------------------
@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (3, 5, 9, 11), (2, 7), (128, 1024), (5, 3, 33, 31),
                                   (4, 32, 56, 56), (2, 64, 112, 112), (7, 8200), (3, 8192), (2, 8196)])
@pytest.mark.parametrize("signed,width", [(False, 8), (True, 8), (False, 4), (True, 2)])
def test_activation_vs_oracle(dev, ops, shape, signed, width):
    rng = np.random.default_rng(hash((shape, signed, width)) % (2 ** 31))
    x = rng.standard_normal(shape).astype(np.float32) * np.float32(rng.uniform(0.01, 30))
    if not signed:
        x = np.where(rng.random(shape) < 0.5, 0, x).astype(np.float32)
    want_y, want_cur, _, want_codes = O.conv_input_fake_quant(x, signed, width)
    y, cur, codes = ops.fake_quant_online(T(x, dev), width, ops.act_flags(signed=signed), want_codes=True)
    assert N(cur)[0] == want_cur
    _eq(N(codes), want_codes.astype(np.int32), "codes")
    _eq(N(y), want_y, "y")
    np.testing.assert_allclose(N(y), want_y, rtol=REL_TOL, atol=0)
    _eq(N(ops.absmax_per_sample(T(x, dev))), O.absmax_per_sample(x), "per-sample abs-max")
    thr = np.float32(want_cur * np.float32(0.7))
    want_y, want_cur, _, want_codes = O.conv_input_fake_quant(x, signed, width, offline_threshold=thr)
    y, cur, codes = ops.fake_quant_offline(T(x, dev), T(np.float32([thr]), dev), width, ops.act_flags(signed=signed),
                                           want_codes=True)
    assert N(cur)[0] == want_cur
    _eq(N(codes), want_codes.astype(np.int32), "offline codes")
    _eq(N(y), want_y, "offline y")


def test_unaligned_views_take_the_scalar_path(dev, ops):
    rng = np.random.default_rng(3)
    big = rng.standard_normal(4 * 1000 + 1).astype(np.float32)
    x = T(big, dev)[1:].reshape(4, 1000)              # 4-byte-aligned only; contiguous
    assert x.data_ptr() % 16 != 0 and x.is_contiguous()
    want_y, want_cur, _, _ = O.conv_input_fake_quant(big[1:].reshape(4, 1000), True, 8)
    y, cur, _ = ops.fake_quant_online(x, 8, ops.act_flags(signed=True))
    assert N(cur)[0] == want_cur
    _eq(N(y), want_y)


def test_act_output_variant_no_abs_no_eps(dev, ops):
    """convert_act.py:49-54: statistic = max (not abs-max), no epsilon, unsigned."""
    rng = np.random.default_rng(11)
    a = rng.standard_normal((6, 4, 5, 5)).astype(np.float32) - np.float32(0.3)
    want_y, want_cur, _, _ = O.act_output_fake_quant(a, 8)
    y, cur, _ = ops.fake_quant_online(T(a, dev), 8, ops.act_flags(no_abs=True, no_eps=True))
    assert N(cur)[0] == want_cur
    _eq(N(y), want_y)
    allneg = -np.abs(a) - 1
    y, cur, _ = ops.fake_quant_online(T(allneg, dev), 8, ops.act_flags(no_abs=True, no_eps=True))
    assert N(cur)[0] == O.act_output_fake_quant(allneg, 8)[1] < 0


def test_batch_mean_is_the_defined_order(dev, ops):
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 127, 128, 1000):
        v = (rng.random(n) * 10).astype(np.float32)
        assert N(ops.batch_mean(T(v, dev)))[0] == O.batch_mean(v)


def test_in_place_apply(dev, ops):
    rng = np.random.default_rng(6)
    x = np.abs(rng.standard_normal((8, 16, 28, 28))).astype(np.float32)
    want, _, _, _ = O.conv_input_fake_quant(x)
    t = T(x, dev)
    y, _, _ = ops.fake_quant_online(t, out=t)
    assert y.data_ptr() == t.data_ptr()
    _eq(N(t), want)


# ---- weights (G5, G6) -------------------------------------------------------------------------------------------
def test_weight_golden(golden, dev, ops):
    g = golden("g5_weight")
    for name in ("dw16", "pw32x16", "c8x4k3"):
        w = g[name + "/w"]
        for qt in ("layer", "group", "channel"):
            for width in (8, 4):
                key = "%s/%s_w%d" % (name, qt, width)
                if key not in g:
                    continue
                rows = 1 if qt == "layer" else (w.shape[0] if qt == "channel" else (16 if name == "dw16" else 1))
                wq, sc = ops.weight_fake_quant(T(w, dev), rows, width, want_scales=True)
                _eq(N(wq), g[key], key)
                _eq(N(sc), O.weight_fake_quant(w, qt, width, num_group=rows)[1], key + " scales")
    for qt in ("layer", "channel"):
        for width in (8, 4):
            w = g["dense/%s_w%d/w" % (qt, width)]
            wq = ops.weight_fake_quant(T(w, dev), 1 if qt == "layer" else w.shape[0], width)
            _eq(N(wq), g["dense/%s_w%d/wq" % (qt, width)], "dense " + qt)


@pytest.mark.parametrize("shape,rows", [((1024, 1, 3, 3), 1024), ((1024, 1, 3, 3), 1), ((512, 512, 3, 3), 512),
                                        ((512, 512, 3, 3), 1), ((1000, 2048), 1000), ((1000, 2048), 1),
                                        ((64, 3, 7, 7), 64), ((5, 9001), 5), ((3, 8193), 3), ((2, 8192), 2)])
def test_weight_vs_oracle_real_layer_shapes(dev, ops, shape, rows):
    rng = np.random.default_rng(sum(shape) + rows)
    w = (rng.standard_normal(shape) * rng.uniform(0.01, 2.0, (shape[0],) + (1,) * (len(shape) - 1))).astype(np.float32)
    qt = "layer" if rows == 1 else "channel"
    for width in (8, 4):
        want, want_sc = O.weight_fake_quant(w, qt, width)
        wq, sc = ops.weight_fake_quant(T(w, dev), rows, width, want_scales=True)
        _eq(N(sc), want_sc, "scales")
        _eq(N(wq), want, "w_q")


@pytest.mark.parametrize("variant", ["F23", "F43", "F63"])
def test_winograd_golden(golden, dev, ops, variant):
    g = golden("g6_winograd")
    G, GI, GTI = ops.winograd_matrices(variant)
    _eq(G, g[variant + "/G"])
    _eq(GI, g[variant + "/GI"], "pinv(G) (LAPACK-dependent; same image on the GPU box)")
    _eq(GTI, g[variant + "/GTI"])
    for name in ("c8x4k3", "dw16"):
        for width in (8, 4):
            w = g["%s/%s_w%d/w" % (variant, name, width)]
            wq = ops.wino_weight_fake_quant(T(w, dev), variant, width, GI=g[variant + "/GI"], GTI=g[variant + "/GTI"])
            _eq(N(wq), g["%s/%s_w%d/wq" % (variant, name, width)], "%s %s w%d" % (variant, name, width))


def test_winograd_resnet_shape_vs_oracle(dev, ops):
    rng = np.random.default_rng(9)
    w = (rng.standard_normal((64, 300, 3, 3)) * 0.05).astype(np.float32)      # cin_g > workgroup size
    want, want_sc, _ = O.wino_weight_fake_quant(w, "F43", 8)
    wq, sc = ops.wino_weight_fake_quant(T(w, dev), "F43", 8, want_scales=True)
    _eq(N(sc), want_sc)
    _eq(N(wq), want)


# ---- EMA (G7) ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["layer_w8", "channel_w4"])
def test_ema_golden(golden, dev, ops, tag):
    g = golden("g7_g9_ema_state")
    cur, ema = g[tag + "/calib_cur"], g[tag + "/calib_ema"]
    state = torch.zeros(cur.shape[1], dtype=torch.float32, device=dev)
    for step in range(cur.shape[0]):
        ops.ema_update(state, T(cur[step], dev), 0.9)
        _eq(N(state), ema[step], "ema step %d" % step)


# ---- histogram / KL (G1-G3: pinned to the real reference module) ---------------------------------------------------
@pytest.mark.parametrize("name", ["halfnormal", "exponential", "relu_outlier", "tiny_range", "shape4d"])
def test_histogram_golden(golden, dev, ops, name):
    g = golden("g1_histogram")
    fm = T(g[name + "/fm"], dev)
    mx = ops.global_max(fm)
    assert N(mx)[0] == g[name + "/max_auto"]
    hist = torch.zeros(2048, dtype=torch.int64, device=dev)
    neg = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.histogram_accumulate(fm, mx, hist, neg)
    _eq(N(ops.hist_to_float(hist)), g[name + "/hist_auto"], "auto")
    assert int(N(neg)[0]) == 0
    hist.zero_()
    ops.histogram_accumulate(fm, T(np.float32([g[name + "/max_fixed"]]), dev), hist)
    _eq(N(hist).astype(np.float32), g[name + "/hist_fixed"], "fixed range")
    h128 = torch.zeros(128, dtype=torch.int64, device=dev)
    ops.histogram_accumulate(fm, mx, h128)
    _eq(N(h128).astype(np.float32), g[name + "/hist_auto_b128"], "128 bins")


def test_histogram_accumulates_across_batches_like_collect_feature_maps(golden, dev, ops):
    g = golden("g3_collect")
    for k in range(3):
        hist = torch.zeros(2048, dtype=torch.int64, device=dev)
        fm_max = None
        for b in g["batches"]:
            fm = np.maximum(b, 0) * np.float32(k + 1)
            if k == 2:
                fm = np.ascontiguousarray(fm[:, :, ::2, ::2])
            t = T(fm, dev)
            if fm_max is None:
                fm_max = ops.global_max(t)
            ops.histogram_accumulate(t, fm_max, hist)
        _eq(N(hist).astype(np.float32), g["hist%d" % k])
        assert N(fm_max)[0] == g["fm_max%d" % k]


def test_histogram_counts_negatives_and_clamps_last_bin(dev, ops):
    x = np.float32([-1.0, 0.0, 0.5, 300.0, 300.0, 1e-30])
    hist = torch.zeros(16, dtype=torch.int64, device=dev)
    neg = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.histogram_accumulate(T(x, dev), T(np.float32([300.0]), dev), hist, neg)
    assert int(N(neg)[0]) == 1
    want, _ = O.discrete_histogram(np.maximum(x, 0), 16, np.float32(300.0))
    _eq(N(hist).astype(np.float32), want)
    assert N(hist)[15] == 2         # fp32(300 + 1e-5) == 300 -> index == bins, clamped (DESIGN.md deviation note)


def test_kl_search_golden_all_cases_one_launch_per_level(golden, dev, ops):
    g = golden("g2_kl")
    names = ["halfnormal", "exponential", "relu_outlier", "accumulated6", "sparse", "spike"]
    hists = T(np.stack([g[n + "/hist"] for n in names]), dev)
    for levels in (256, 128, 16, 8):
        best = N(ops.kl_search(hists, levels, levels))
        want = np.asarray([int(g["%s/best_L%d" % (n, levels)]) for n in names])
        _eq(best, want.astype(np.int32), "levels=%d" % levels)
    for n in ("halfnormal", "sparse"):
        h = T(g[n + "/hist_b256"], dev)
        for levels in (16, 32):
            assert int(N(ops.kl_search(h, levels, levels))[0]) == int(g["%s/best_b256_L%d" % (n, levels)])


# ---- int-code path (G8) ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["u01", "normal", "shifted"])
@pytest.mark.parametrize("t", ["int8", "uint8"])
def test_quantize_codes_golden(golden, dev, ops, name, t):
    g = golden("g8_quantized_conv")
    codes, rng_dev = ops.quantize_codes(T(g[name + "/x"], dev), t)
    _eq(N(codes), g["%s/%s_codes" % (name, t)])
    assert N(rng_dev)[2] == g["%s/%s_scale" % (name, t)]
    _eq(N(ops.dequantize(codes, rng_dev[2:3])), g["%s/%s_deq" % (name, t)])


def test_quantize_codes_fixed_range_and_explicit_scale(dev, ops):
    rng = np.random.default_rng(2)
    x = rng.standard_normal(1000).astype(np.float32) * 3
    want, sc = O.quantize_codes(x, fixed_range=(-2.0, 2.0))
    codes, r = ops.quantize_codes(T(x, dev), "range", T(np.float32([-2, 2, 0]), dev))
    _eq(N(codes), want)
    assert N(r)[2] == sc
    codes, _ = ops.quantize_codes(T(x, dev), "scale", T(np.float32([-1e9, 1e9, 0.0123]), dev))
    _eq(N(codes), O.roundf((x / np.float32(0.0123)).astype(np.float32)).astype(np.int32))


# ---- full-size properties (BASELINE configs[1]: mobilenet1.0 activations at batch 128) ---------------------------------
def test_headline_tensor_properties(dev, ops):
    """(128, 64, 112, 112) fp32 = 411 MB, the largest MobileNet activation.  Size-independent checks:
    codes in range; y == codes*scale; idempotence (re-quantising with the same threshold is the identity);
    sampled slices equal the oracle."""
    torch.manual_seed(7)
    x = torch.relu(torch.randn(128, 64, 112, 112, device=dev)) * 1.7
    y, cur, codes = ops.fake_quant_online(x, 8, 0, want_codes=True)
    cur_h = N(cur)[0]
    want_cur = O.batch_mean(N(torch.amax(x.abs().reshape(128, -1), dim=1)))
    assert cur_h == want_cur
    assert int(codes.min()) == 0 and int(codes.max()) == 255
    scale = np.float32(cur_h / np.float32(255))
    assert torch.equal(y, codes.float() * float(scale))
    y2, _, codes2 = ops.fake_quant_offline(y, cur, 8, 0, want_stat=False, want_codes=True)
    assert torch.equal(codes2, codes) and torch.equal(y2, y)
    for n in (0, 63, 127):
        xs = N(x[n, 5])
        want = O.ste_forward(xs, scale, cur_h, 0.0)
        _eq(N(y[n, 5]), want, "slice %d" % n)
    del y2, codes2, codes
    hist = torch.zeros(2048, dtype=torch.int64, device=dev)
    mx = ops.global_max(x)
    ops.histogram_accumulate(x, mx, hist)
    assert int(hist.sum()) == int((x != 0).sum())      # every non-zero lands in exactly one bin


# ---- fused producer: BN + activation + statistic (quantize/fuse.py) -------------------------------------------------
@pytest.mark.parametrize("shape", [(4, 8, 7, 7), (3, 5, 9, 11), (2, 1024, 7, 7), (8, 32, 56, 56), (2, 64, 112, 112),
                                   (5, 3, 1, 1), (6, 10)])
@pytest.mark.parametrize("act", ["relu", "relu6", "none"])
def test_bn_act_stat_vs_oracle(dev, ops, shape, act):
    rng = np.random.default_rng(sum(shape))
    x = (rng.standard_normal(shape) * 3).astype(np.float32)
    c = shape[1]
    scale = rng.uniform(0.2, 2.0, c).astype(np.float32) * rng.choice([-1, 1], c).astype(np.float32)
    shift = rng.standard_normal(c).astype(np.float32)
    y, stat = ops.bn_act_stat(T(x, dev), T(scale, dev), T(shift, dev), act)
    want = O.bn_act(x, scale, shift, act)
    _eq(N(y), want, "bn_act")
    _eq(N(stat), O.absmax_per_sample(want), "per-sample statistic of the output")
    y2, none = ops.bn_act_stat(T(x, dev), T(scale, dev), T(shift, dev), act, want_stat=False)
    assert none is None
    _eq(N(y2), want)
    # the consumer's apply-only online path == the two-pass online path
    a, cur_a, codes_a = ops.fake_quant_online(y, 8, 0, want_codes=True)
    b, cur_b, codes_b = ops.fake_quant_online_prestat(y, stat, 8, 0, want_codes=True)
    assert N(cur_a)[0] == N(cur_b)[0]
    _eq(N(codes_a), N(codes_b))
    _eq(N(a), N(b))


# ---- fused depthwise 3x3: quantise-on-load + BN/act/statistic epilogue ------------------------------------------------
DW_SHAPES = [(2, 8, 7, 7), (3, 16, 14, 14), (2, 32, 28, 28), (2, 8, 56, 56), (2, 4, 112, 112), (1, 3, 9, 11),
             (2, 5, 13, 6), (4, 1024, 7, 7), (2, 6, 70, 70), (1, 2, 113, 113),
             (2, 6, 14, 20), (3, 5, 7, 10), (1, 3, 14, 62), (5, 4, 7, 3), (70, 33, 14, 14), (37, 20, 7, 7)]   # whole-plane form (K2o)


@pytest.mark.parametrize("shape", DW_SHAPES)
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("mode", ["plain", "online", "offline_signed", "bn_relu_online", "bias_relu6"])
def test_dwconv3x3_vs_oracle(dev, ops, shape, stride, mode):
    rng = np.random.default_rng(sum(shape) * 7 + stride)
    n, c, h, w = shape
    x = (rng.standard_normal(shape) * 2).astype(np.float32)
    if "signed" not in mode:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((c, 1, 3, 3)) * 0.5).astype(np.float32)
    kw, okw = {}, {}
    if mode in ("online", "bn_relu_online"):
        stat = O.absmax_per_sample(x)
        kw.update(in_stat=T(stat, dev), width=8, flags=0)
        okw.update(in_max=O.batch_mean(stat), signed=False, width=8)
    if mode == "offline_signed":
        thr = np.float32(1.7)
        kw.update(in_thr=T(np.float32([thr]), dev), width=4, flags=ops.act_flags(signed=True))
        okw.update(in_max=thr, signed=True, width=4)
    if mode == "bn_relu_online":
        sc = rng.uniform(0.3, 1.5, c).astype(np.float32)
        sh = rng.standard_normal(c).astype(np.float32)
        kw.update(bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu")
        okw.update(bn_scale=sc, bn_shift=sh, act="relu")
    if mode == "bias_relu6":
        b = rng.standard_normal(c).astype(np.float32)
        kw.update(bias=T(b, dev), act="relu6")
        okw.update(bias=b, act="relu6")
    cur = torch.zeros(1, device=dev)
    y, stat_out = ops.dwconv3x3(T(x, dev), T(wt, dev), stride=stride, cur_out=cur, **kw)
    want = O.dwconv3x3(x, wt, stride=stride, **okw)
    got = N(y)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    assert (got != want).mean() < 1e-3                         # fmaf emulation differs only at double-rounding ties
    _eq(N(stat_out), O.absmax_per_sample(got), "statistic of the produced output")
    if "online" in mode:
        assert N(cur)[0] == okw["in_max"]
    # against an independent implementation (torch conv on the quantised input), loose
    import torch.nn.functional as TF
    xq = x if "in_max" not in okw else O.conv_input_fake_quant(x, okw["signed"], okw["width"],
                                                                offline_threshold=okw["in_max"])[0]
    ref = TF.conv2d(torch.from_numpy(xq).double(), torch.from_numpy(wt).double(), None, stride=stride, padding=1,
                    groups=c).numpy()
    if mode == "plain":
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


# ---- residual tail: (a + b).relu() + statistic ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(4, 256, 56, 56), (3, 64, 7, 7), (2, 10, 5, 3), (5, 2048, 7, 7), (1, 16, 32, 32)])
@pytest.mark.parametrize("act", ["relu", "relu6", "none"])
def test_add_act_stat_vs_oracle(dev, ops, shape, act):
    from oracle import host as H
    rng = np.random.default_rng(sum(shape))
    a = (rng.standard_normal(shape) * 3).astype(np.float32)
    b = (rng.standard_normal(shape) * 3).astype(np.float32)
    y, stat = ops.add_act_stat(T(a, dev), T(b, dev), act)
    want, wstat = H.add_act(a, b, act, want_stat=True)
    _eq(N(y), want, "a + b, activation")
    _eq(N(stat), wstat, "statistic")
    ref = a + b
    ref = np.maximum(ref, 0) if act != "none" else ref
    ref = np.minimum(ref, 6) if act == "relu6" else ref
    _eq(want, ref.astype(np.float32), "host twin vs numpy")
    y2, none = ops.add_act_stat(T(a, dev)[:, :, 1:].contiguous(), T(b, dev)[:, :, 1:].contiguous(), act, want_stat=False)
    assert none is None
    _eq(N(y2), want[:, :, 1:], "unaligned / no statistic")


@pytest.mark.parametrize("shape", [(2, 64, 56, 56), (3, 5, 9, 12), (1, 3, 8, 8), (2, 7, 7, 5), (130, 2, 6, 8)])
@pytest.mark.parametrize("act", ["relu", "none", "relu6"])
def test_bn_add_act_stat_is_the_two_passes_in_one(dev, ops, shape, act):
    """fq_bn_add_act_stat (round 6): BatchNorm, the residual add, the activation and the statistic in one pass - bit for bit
    fq_bn_act_stat without activation followed by fq_add_act_stat, and the host twin; aligned and unaligned inputs; with a
    histogram sink the counts the separate histogram pass adds."""
    from oracle import host as H
    rng = np.random.default_rng(sum(shape) + 1)
    x = (rng.standard_normal(shape) * 3).astype(np.float32)
    r = (rng.standard_normal(shape) * 2).astype(np.float32)
    sc = (0.5 + rng.random(shape[1])).astype(np.float32) * np.where(rng.random(shape[1]) < 0.2, -1, 1).astype(np.float32)
    sh = rng.standard_normal(shape[1]).astype(np.float32)
    y, stat = ops.bn_act_stat(T(x, dev), T(sc, dev), T(sh, dev), act, residual=T(r, dev))
    t, _ = ops.bn_act_stat(T(x, dev), T(sc, dev), T(sh, dev), "none", want_stat=False)
    y2, stat2 = ops.add_act_stat(t, T(r, dev), act)
    _eq(N(y), N(y2), "one pass vs two")
    _eq(N(stat), N(stat2), "statistic")
    want, wstat = H.add_act(H.bn_act(x, sc, sh, "none"), r, act, want_stat=True)
    _eq(N(y), want, "host twins")
    _eq(N(stat), wstat, "host statistic")
    xu, ru = T(x, dev).reshape(-1)[1:].clone(), T(r, dev).reshape(-1)[1:].clone()      # storage not 16-byte aligned
    if shape[0] > 1:
        xs, rs = T(x, dev)[1:], T(r, dev)[1:]
        y3, stat3 = ops.bn_act_stat(xs.contiguous(), T(sc, dev), T(sh, dev), act, residual=rs.contiguous())
        _eq(N(y3), want[1:], "a batch slice")
    del xu, ru


@pytest.mark.parametrize("shape", [(2, 64, 112, 112), (3, 5, 9, 12), (1, 3, 8, 8), (2, 4, 7, 4), (130, 2, 6, 8)])
@pytest.mark.parametrize("act", ["relu", "none", "relu6"])
def test_bn_act_maxpool_stat_vs_oracle(dev, ops, shape, act):
    """BatchNorm -> activation -> MaxPool2D(3, 2, 1) in one pass (the head of the ImageNet ResNets): odd and even heights,
    negative BatchNorm scales (the maximum must be taken AFTER the affine map), more samples than workgroups per sample."""
    from oracle import host as H
    rng = np.random.default_rng(sum(shape) + len(act))
    x = (rng.standard_normal(shape) * 3).astype(np.float32)
    sc = rng.standard_normal(shape[1]).astype(np.float32)
    sh = rng.standard_normal(shape[1]).astype(np.float32)
    y, stat = ops.bn_act_maxpool_stat(T(x, dev), T(sc, dev), T(sh, dev), act)
    want = O.bn_act_maxpool(x, sc, sh, act)
    _eq(N(y), want, "BN + act + max pooling")
    _eq(N(stat), O.absmax_per_sample(want), "statistic")
    _eq(H.bn_act_maxpool(x, sc, sh, act), want, "host twin vs numpy oracle")
    ref = torch.nn.functional.max_pool2d(torch.from_numpy(O.bn_act(x, sc, sh, act)), 3, 2, 1).numpy()
    _eq(want, ref, "oracle vs torch's pooling of the oracle's BN + act")


# ---- pointwise convolution on integer codes (int8 MFMA) ---------------------------------------------------------------
PW_CASES = [  # (n, cin, cout, h, w)
    (2, 32, 64, 28, 28), (3, 64, 128, 14, 14), (2, 128, 128, 9, 12), (2, 128, 256, 14, 14), (5, 512, 512, 7, 7),
    (3, 1024, 1024, 7, 7), (2, 24, 40, 5, 7), (1, 3, 8, 4, 4), (2, 96, 576, 6, 6), (4, 1024, 1000, 1, 1),
    (2, 960, 320, 7, 7), (2, 144, 24, 14, 14), (2, 16, 96, 9, 9),
    (3, 256, 512, 5, 6)]     # tile form, K = 256 instantiation, ragged last 32-pixel tile (90 columns)


@pytest.mark.parametrize("case", PW_CASES, ids=["%dx%d->%d@%dx%d" % c for c in PW_CASES])
@pytest.mark.parametrize("mode", ["online_u8_layer", "offline_s8_channel_w4", "online_u8_bn_relu", "dense_quirk_bias"])
def test_pwconv_i8_vs_oracle(dev, ops, case, mode):
    _pwconv_case(dev, ops, case, mode)


@pytest.mark.parametrize("case,mode", [((84, 512, 512, 14, 14), "online_u8_bn_relu"),
                                       ((86, 256, 512, 14, 14), "offline_s8_channel_w4")],
                         ids=["512->512@14x14x84", "256->512@14x14x86"])
def test_pwconv_i8_many_tiles_vs_oracle(dev, ops, case, mode):
    """Hundreds of 32-column tiles (several resident rounds of the split form's workgroups, XCD-contiguous work order);
    86 * 196 columns also end in a ragged tile."""
    _pwconv_case(dev, ops, case, mode)


# Every form of fq_pwconv_i8 named explicitly (FQ_PW_FORM bits of the call), on shapes it accepts — the shape-based choice
# above reaches: split (every shape whose padded K/32 is instantiated and that has <= 4096 tiles or that the streaming form
# cannot take), stream (K <= 256 with the whole weight matrix in LDS, large planes), two_kernels (everything else).
FORM_CASES = [
    ("two_kernels", (2, 24, 40, 5, 7)), ("two_kernels", (2, 512, 512, 7, 7)), ("two_kernels", (2, 960, 320, 7, 7)),
    ("two_kernels", (2, 144, 24, 14, 14)), ("two_kernels", (2, 448, 96, 6, 6)),
    ("stream", (2, 32, 64, 28, 28)), ("stream", (2, 128, 256, 14, 14)), ("stream", (3, 256, 256, 9, 7)),
    # split: K/32 = 8 / 16 / 32 / 64; one, two and four channel tiles per wavefront (Cout 128 / 256 / 512+); a ragged last
    # tile; tiles that span many samples (1x1 planes: more samples than statistic slots)
    ("split", (3, 1024, 1024, 7, 7)), ("split", (3, 256, 512, 5, 6)), ("split", (5, 512, 1024, 7, 7)),
    ("split", (2, 512, 512, 14, 14)), ("split", (2, 256, 128, 9, 7)), ("split", (2, 2048, 512, 7, 7)),
    ("split", (2, 512, 256, 7, 7)), ("split", (70, 512, 512, 1, 1)),
    # ... padded K (Cin % 32 != 0: 24 -> 64, 144 -> 192, 960 -> 960 = 30 slabs, 3 -> 64), partial and missing channel tiles
    # (Cout 40, 24, 320, 8, 1000), every remaining K/32 instantiation (2, 4, 6, 10, 12, 18, 30)
    ("split", (2, 24, 40, 5, 7)), ("split", (2, 96, 576, 6, 6)), ("split", (2, 144, 24, 14, 14)),
    ("split", (2, 960, 320, 7, 7)), ("split", (2, 16, 96, 9, 9)), ("split", (1, 3, 8, 4, 4)),
    ("split", (2, 384, 64, 14, 14)), ("split", (2, 576, 160, 7, 7)), ("split", (2, 320, 1280, 7, 7)),
    ("split", (4, 1024, 1000, 1, 1)),
    # sample (14x14 planes, half a sample per workgroup): K/32 = 8 and 16, one and two channel groups of 512, a sample count that
    # leaves XCD shares ragged
    ("sample", (3, 512, 512, 14, 14)), ("sample", (9, 256, 512, 14, 14)), ("sample", (17, 256, 1024, 14, 14)),
    # ... 256 channels per workgroup, 28x28 planes in seven blocks of 112 pixels, one block of 100, K/32 = 4
    ("sample", (2, 512, 256, 14, 14)), ("sample", (3, 128, 256, 28, 28)), ("sample", (2, 256, 256, 28, 28)),
    ("sample", (5, 128, 512, 10, 10)), ("sample", (2, 256, 768, 16, 24)),
    # ... whole small planes in two pixel tiles: 7x7 (49 pixels: the last one has a load of its own) and 8x8, K/32 = 16 and 32
    ("sample", (5, 512, 1024, 7, 7)), ("sample", (3, 1024, 1024, 7, 7)), ("sample", (2, 512, 256, 8, 8)),
    ("sample", (9, 1024, 512, 7, 7)), ("sample", (3, 1024, 256, 14, 14)), ("sample", (3, 2048, 512, 7, 7)),
    ("sample", (2, 2048, 256, 8, 8)),
    # pipe (round 5: weights resident in registers, sub-tiles of <= 7 / 8 pixel groups through LDS-DMA): K = 512 and 256; 14x14
    # in two blocks of 25 + 24 groups (four sub-tiles each), two channel groups, XCD shares left ragged by the sample count,
    # one block of 25 groups (10x10), three blocks of 32 (16x24), a single sub-tile (4x4), two sub-tiles of 5 + 4 (6x6)
    ("pipe", (3, 512, 512, 14, 14)), ("pipe", (9, 256, 512, 14, 14)), ("pipe", (17, 256, 1024, 14, 14)),
    ("pipe", (2, 512, 1024, 10, 10)), ("pipe", (2, 256, 512, 16, 24)), ("pipe", (2, 512, 512, 4, 4)),
    ("pipe", (5, 512, 512, 6, 6)), ("pipe", (2, 512, 512, 28, 28)),
    # rows (planes of one pixel = the classifier): the model zoo's heads at batch 128, a partial sample tile, a partial unit
    # tile with padded K (100 -> 128), K in two rounds of slabs (2048), fewer slabs than wavefronts (64)
    ("rows", (128, 1024, 1000, 1, 1)), ("rows", (70, 512, 512, 1, 1)), ("rows", (33, 100, 37, 1, 1)),
    ("rows", (1, 2048, 1000, 1, 1)), ("rows", (3, 64, 10, 1, 1)), ("rows", (40, 1280, 1000, 1, 1))]


def _pipe_built():
    try:
        from quantization.mxnet_amd import _lib
        return bool(_lib.LIB.fq_build_has(b"pipe"))
    except Exception:
        return False


# (the pipe form is shelved - DESIGN.md 3.3 - and compiled by `csrc/build.py --dev` only: its cases run against such a library,
# FQ_LIB_PATH=.../libfakequant_dev.so)
if not _pipe_built():
    FORM_CASES = [fc for fc in FORM_CASES if fc[0] != "pipe"]


@pytest.mark.parametrize("form,case", FORM_CASES, ids=["%s-%dx%d->%d@%dx%d" % ((f,) + c) for f, c in FORM_CASES])
@pytest.mark.parametrize("mode", ["online_u8_bn_relu", "offline_s8_channel_w4", "dense_quirk_bias"])
def test_pwconv_i8_every_form_vs_oracle(dev, ops, form, case, mode):
    _pwconv_case(dev, ops, case, mode, form=form)


def _pwconv_case(dev, ops, case, mode, form=None):
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case))
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    if "s8" not in mode:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * rng.uniform(0.05, 1.0, (cout, 1, 1, 1))).astype(np.float32)
    per_channel = "channel" in mode
    wt_width = 4 if "w4" in mode else 8
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), 1 if per_channel else cout, wt_width)
    ocodes, oscales = O.weight_codes(wt, 1 if per_channel else cout, wt_width)
    _eq(N(codes)[:cout, :cin], ocodes.astype(np.int8), "weight codes")
    assert not N(codes)[cout:].any() and not N(codes)[:, cin:].any()
    _eq(N(scales), oscales, "weight scales")
    _eq(N(rowsum), ocodes.sum(axis=1).astype(np.int32), "row sums")
    kw, okw = {}, {}
    if mode.startswith("online"):
        stat = O.absmax_per_sample(x)
        kw.update(in_stat=T(stat, dev), width=8, flags=0)
        okw.update(in_max=O.batch_mean(stat), signed=False, width=8)
    elif mode.startswith("offline"):
        thr = np.float32(2.3)
        kw.update(in_thr=T(np.float32([thr]), dev), width=8, flags=ops.act_flags(signed=True))
        okw.update(in_max=thr, signed=True, width=8)
    else:   # Dense quirk: signed levels, clip at zero
        stat = O.absmax_per_sample(x)
        kw.update(in_stat=T(stat, dev), width=8, flags=ops.act_flags(signed=True, lo_neg_max=False))
        okw.update(in_max=O.batch_mean(stat), signed=True, width=8, lo_neg_max=False)
    if "bn_relu" in mode:
        sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
        sh = rng.standard_normal(cout).astype(np.float32)
        kw.update(bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu")
        okw.update(bn_scale=sc, bn_shift=sh, act="relu")
    if "bias" in mode:
        b = rng.standard_normal(cout).astype(np.float32)
        kw.update(bias=T(b, dev))
        okw.update(bias=b)
    cur = torch.zeros(1, device=dev)
    y, stat_out = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, cur_out=cur, form=form, **kw)
    want = O.pwconv_i8(x, wt, 1 if per_channel else cout, wt_width, **okw)
    got = N(y)
    _eq(got, want, "pointwise int8 convolution (exact integer sums)")
    _eq(N(stat_out), O.absmax_per_sample(got), "statistic")
    # the same layer through the reference's formulation (fp32 conv of the dequantised tensors): equal up to fp32 conv noise
    wq = O.weight_fake_quant(wt, "channel" if per_channel else "layer", wt_width)[0]
    xq = O.ste_forward(x, O.act_scale(okw["in_max"], okw["signed"], okw["width"]), okw["in_max"],
                       -okw["in_max"] if okw.get("lo_neg_max", okw["signed"]) else 0.0)
    ref = np.einsum("oc,nchw->nohw", wq.reshape(cout, cin).astype(np.float64), xq.astype(np.float64))
    if "bias" in mode:
        ref = ref + okw["bias"].reshape(1, -1, 1, 1)
    if "bn_relu" not in mode:
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


PW_RES_CASES = [("split", (2, 256, 1024, 14, 14)), ("split", (3, 512, 2048, 7, 7)), ("stream", (2, 64, 256, 28, 28)),
                ("split", (2, 144, 24, 14, 14)), ("split", (3, 384, 64, 7, 7)), (None, (2, 128, 512, 9, 11)),
                ("sample", (3, 256, 1024, 14, 14)), ("sample", (2, 128, 512, 28, 28)), ("sample", (2, 512, 512, 14, 14)),
                ("sample", (3, 512, 2048, 7, 7)), (None, (9, 512, 2048, 7, 7)), ("sample", (2, 512, 1024, 8, 8))]


@pytest.mark.parametrize("form,case", PW_RES_CASES, ids=["%s-%dx%d->%d@%dx%d" % ((f,) + c) for f, c in PW_RES_CASES])
@pytest.mark.parametrize("mode", ["online_u8_bn_relu", "offline_s8_channel_w4_bn_none"])
def test_pwconv_i8_residual_vs_oracle(dev, ops, form, case, mode):
    """The tail of a residual unit in the epilogue of its last 1x1 convolution: y = act(BN(conv) + shortcut) (ResNet: ReLU,
    MobileNetV2: no activation); partial channel tiles (24, 64 channels) included."""
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + 11)
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    if "s8" not in mode:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * rng.uniform(0.05, 1.0, (cout, 1, 1, 1))).astype(np.float32)
    res = (rng.standard_normal((n, cout, h, w)) * 3).astype(np.float32)
    per_channel = "channel" in mode
    wt_width = 4 if "w4" in mode else 8
    rps = 1 if per_channel else cout
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), rps, wt_width)
    kw, okw = {}, {}
    if mode.startswith("online"):
        stat = O.absmax_per_sample(x)
        kw.update(in_stat=T(stat, dev), width=8, flags=0)
        okw.update(in_max=O.batch_mean(stat), signed=False, width=8)
    else:
        thr = np.float32(2.3)
        kw.update(in_thr=T(np.float32([thr]), dev), width=8, flags=ops.act_flags(signed=True))
        okw.update(in_max=thr, signed=True, width=8)
    act = "relu" if "relu" in mode else None
    sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    y, stat_out = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, cur_out=torch.zeros(1, device=dev), bn_scale=T(sc, dev),
                                bn_shift=T(sh, dev), act=act, residual=T(res, dev), form=form, **kw)
    want = O.pwconv_i8(x, wt, rps, wt_width, bn_scale=sc, bn_shift=sh, act=act, residual=res, **okw)
    _eq(N(y), want, "pointwise int8 convolution with a residual operand")
    _eq(N(stat_out), O.absmax_per_sample(want), "statistic")
    # ... which is what the separate passes compute
    plain = O.pwconv_i8(x, wt, rps, wt_width, bn_scale=sc, bn_shift=sh, act=None, **okw)
    two_pass = (plain + res).astype(np.float32)
    _eq(want, np.maximum(two_pass, 0) if act == "relu" else two_pass, "oracle: fused tail == convolution, then add, then act")
    from oracle import host as H
    _eq(H.pwconv_i8(x, wt, rps, wt_width, bn_scale=sc, bn_shift=sh, act=act, residual=res, **okw), want, "host twin")


@pytest.mark.parametrize("case", [(3, 144, 24, 14, 14), (4, 192, 40, 7, 7), (3, 144, 24, 56, 56)],
                         ids=lambda c: "%dx%d->%d@%dx%d" % c)
def test_pwconv_i8_residual_partial_tile_statistic_ignores_the_next_sample(dev, ops, case):
    """A partial channel tile (Cout % 32 != 0) with a residual operand that DWARFS the convolution: the lanes
    of the channels past Cout must not pick up the next sample's residual (the buffer resource runs to the end of the
    tensor) - it would not be stored, but it would win the per-sample statistic.  Sample 0's statistic is the tell-tale:
    its masked lanes would read sample 1, whose residual is made 1000x larger here."""
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + 3)
    x = np.maximum(rng.standard_normal((n, cin, h, w)), 0).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.01).astype(np.float32)
    res = (rng.standard_normal((n, cout, h, w)) * 100).astype(np.float32)
    res[1:] *= 1000.0
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), cout, 8)
    stat = O.absmax_per_sample(x)
    sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    for form in ("split", None):
        y, stat_out = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, cur_out=torch.zeros(1, device=dev),
                                    bn_scale=T(sc, dev), bn_shift=T(sh, dev), act=None, residual=T(res, dev), form=form,
                                    in_stat=T(stat, dev), width=8, flags=0)
        want = O.pwconv_i8(x, wt, cout, 8, bn_scale=sc, bn_shift=sh, act=None, residual=res,
                           in_max=O.batch_mean(stat), signed=False, width=8)
        _eq(N(y), want, "output (form %s)" % form)
        _eq(N(stat_out), O.absmax_per_sample(want), "per-sample statistic (form %s)" % form)


PW_S2_CASES = [(2, 256, 512, 14, 14), (3, 512, 1024, 7, 7), (2, 64, 128, 9, 11), (2, 1024, 2048, 5, 5), (3, 256, 128, 28, 28),
               (4, 96, 40, 6, 7)]


@pytest.mark.parametrize("case", PW_S2_CASES, ids=["%dx%d->%d@%dx%d" % c for c in PW_S2_CASES])
@pytest.mark.parametrize("mode", ["online_u8_bn_relu", "offline_s8_channel_w4"])
def test_pwconv_i8_stride2_vs_oracle(dev, ops, case, mode):
    """The shortcut / first 1x1 convolutions of the ResNet stages: stride 2, no padding, odd and even planes.  The batch
    statistic is the one of the WHOLE input (what the reference's fake-quant in front of the convolution sees)."""
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + 3)
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    if "s8" not in mode:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin, 1, 1)) * rng.uniform(0.05, 1.0, (cout, 1, 1, 1))).astype(np.float32)
    per_channel = "channel" in mode
    wt_width = 4 if "w4" in mode else 8
    rps = 1 if per_channel else cout
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), rps, wt_width)
    kw, okw = {}, {}
    if mode.startswith("online"):
        stat = O.absmax_per_sample(x)
        kw.update(in_stat=T(stat, dev), width=8, flags=0)
        okw.update(in_max=O.batch_mean(stat), signed=False, width=8)
    else:
        thr = np.float32(2.3)
        kw.update(in_thr=T(np.float32([thr]), dev), width=8, flags=ops.act_flags(signed=True))
        okw.update(in_max=thr, signed=True, width=8)
    if "bn_relu" in mode:
        sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
        sh = rng.standard_normal(cout).astype(np.float32)
        kw.update(bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu")
        okw.update(bn_scale=sc, bn_shift=sh, act="relu")
    y, stat_out = ops.pwconv_i8(T(x, dev), codes, scales, rowsum, cur_out=torch.zeros(1, device=dev), stride=2, **kw)
    want = O.pwconv_i8(x, wt, rps, wt_width, stride=2, **okw)
    assert tuple(y.shape) == want.shape == (n, cout, (h + 1) // 2, (w + 1) // 2)
    _eq(N(y), want, "strided pointwise int8 convolution")
    _eq(N(stat_out), O.absmax_per_sample(want), "statistic")
    from oracle import host as H
    _eq(H.pwconv_i8(x, wt, rps, wt_width, stride=2, **okw), want, "host twin vs numpy oracle")
    with pytest.raises(Exception):     # a strided call on a shape only the generic form would take
        ops.pwconv_i8(torch.zeros(1, 448, 4, 4, device=dev), *ops.weight_codes(torch.ones(32, 448, 1, 1, device=dev), 32, 8),
                      in_thr=T(np.float32([1.0]), dev), stride=2)


def test_offline_consumers_report_the_batch_statistic_on_the_side(dev, ops):
    """Offline mode with the per-sample maxima passed as well: the stored threshold quantises, `current_input_max` (which the
    reference computes in every mode, convert_conv2d.py:56) comes out of the same launch - for every fused consumer."""
    rng = np.random.default_rng(41)
    thr = T(np.float32([1.7]), dev)

    def check(run, x):
        stat = O.absmax_per_sample(x)
        cur = torch.full((1,), -1.0, device=dev)
        y_both, s_both = run(dict(in_thr=thr, in_stat=T(stat, dev), cur_out=cur))
        y_off, s_off = run(dict(in_thr=thr))
        assert torch.equal(y_both, y_off) and torch.equal(s_both, s_off)
        _eq(N(cur), np.float32([O.batch_mean(stat)]), "current_input_max")

    for n, cin, cout, h, w, form in [(3, 512, 256, 7, 7, "split"), (2, 64, 128, 28, 28, "stream"), (2, 448, 96, 6, 6, "two_kernels")]:
        x = np.maximum(rng.standard_normal((n, cin, h, w)) * 2, 0).astype(np.float32)
        wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.2).astype(np.float32)
        codes = ops.weight_codes(T(wt, dev), cout, 8)
        check(lambda kw: ops.pwconv_i8(T(x, dev), *codes, width=8, flags=0, act="relu", form=form, **kw), x)
    for n, c, h, w, stride in [(2, 64, 112, 112, 1), (3, 128, 14, 14, 2), (3, 256, 7, 7, 1)]:      # the three depthwise forms
        x = np.maximum(rng.standard_normal((n, c, h, w)) * 2, 0).astype(np.float32)
        wt = (rng.standard_normal((c, 1, 3, 3)) * 0.3).astype(np.float32)
        check(lambda kw: ops.dwconv3x3(T(x, dev), T(wt, dev), stride=stride, width=8, flags=0, act="relu", **kw), x)
    x = np.maximum(rng.standard_normal((2, 64, 9, 9)) * 2, 0).astype(np.float32)
    wt = (rng.standard_normal((64, 64, 3, 3)) * 0.2).astype(np.float32)
    codes = ops.weight_codes_3x3(T(wt, dev), 64, 8)
    check(lambda kw: ops.conv3x3_i8(T(x, dev), *codes, width=8, flags=0, act="relu", **kw), x)


# ---- dense 3x3 convolution on integer codes (int8 MFMA, implicit GEMM over (tap, ci)) -----------------------------------
C3_CASES = [  # (n, cin, cout, h, w): every K/32 (2, 4, 8, 16), both wavefront arrangements (Cout < 128 / >= 128), partial
    # channel tiles (Cout 96, 160), planes narrower than / as wide as / wider than a pixel tile, ragged last blocks, blocks
    # that span several samples (3x3 and 7x7 planes)
    (2, 64, 64, 9, 11), (3, 128, 128, 7, 7), (2, 256, 256, 5, 6), (1, 512, 512, 7, 7), (2, 64, 128, 14, 14),
    (5, 64, 96, 3, 3), (2, 128, 160, 28, 28), (1, 64, 64, 56, 56), (9, 256, 32, 4, 4), (1, 128, 64, 1, 50)]


@pytest.mark.parametrize("case", C3_CASES, ids=["%dx%d->%d@%dx%d" % c for c in C3_CASES])
@pytest.mark.parametrize("mode", ["online_u8_bn_relu", "offline_s8_channel_w4", "online_s8_bias"])
def test_conv3x3_i8_vs_oracle(dev, ops, case, mode):
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + 7)
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    if "s8" not in mode:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * rng.uniform(0.05, 1.0, (cout, 1, 1, 1))).astype(np.float32)
    per_channel = "channel" in mode
    wt_width = 4 if "w4" in mode else 8
    rps = 1 if per_channel else cout
    codes, scales, rowsum = ops.weight_codes_3x3(T(wt, dev), rps, wt_width)
    ocodes, oscales = O.weight_codes(wt, rps, wt_width)
    _eq(N(scales), oscales, "weight scales")
    _eq(N(rowsum), ocodes.sum(axis=1).astype(np.int32), "row sums")
    _eq(N(codes)[:cout], ocodes.reshape(cout, cin, 3, 3).transpose(0, 2, 3, 1).reshape(cout, -1).astype(np.int8),
        "weight codes in (tap, ci) order")
    kw, okw = {}, {}
    signed = "s8" in mode
    if mode.startswith("online"):
        stat = O.absmax_per_sample(x)
        kw.update(in_stat=T(stat, dev), width=8, flags=ops.act_flags(signed=signed))
        okw.update(in_max=O.batch_mean(stat), signed=signed, width=8)
    else:
        thr = np.float32(2.3)
        kw.update(in_thr=T(np.float32([thr]), dev), width=8, flags=ops.act_flags(signed=True))
        okw.update(in_max=thr, signed=True, width=8)
    if "bn_relu" in mode:
        sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
        sh = rng.standard_normal(cout).astype(np.float32)
        kw.update(bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu")
        okw.update(bn_scale=sc, bn_shift=sh, act="relu")
    if "bias" in mode:
        b = rng.standard_normal(cout).astype(np.float32)
        kw.update(bias=T(b, dev))
        okw.update(bias=b)
    cur = torch.zeros(1, device=dev)
    y, stat_out = ops.conv3x3_i8(T(x, dev), codes, scales, rowsum, cur_out=cur, **kw)
    want = O.conv3x3_i8(x, wt, rps, wt_width, **okw)
    got = N(y)
    _eq(got, want, "dense 3x3 int8 convolution (exact integer sums)")
    _eq(N(stat_out), O.absmax_per_sample(got), "statistic")
    if mode.startswith("online"):
        _eq(N(cur), np.float32([okw["in_max"]]), "current_input_max")
    # the host twin (full-size oracle of the net tests) agrees as well
    from oracle import host as H
    hy = H.conv3x3_i8(x, wt, rps, wt_width, **okw)
    _eq(hy, want, "host twin vs numpy oracle")
    # the reference's formulation: fp32 convolution of the two dequantised tensors, equal up to fp32 summation noise
    wq = O.weight_fake_quant(wt, "channel" if per_channel else "layer", wt_width)[0]
    xq = O.ste_forward(x, O.act_scale(okw["in_max"], okw["signed"], okw["width"]), okw["in_max"],
                       -okw["in_max"] if okw["signed"] else 0.0)
    ref = torch.nn.functional.conv2d(torch.from_numpy(xq.astype(np.float64)), torch.from_numpy(wq.astype(np.float64)),
                                     padding=1).numpy()
    if "bias" in mode:
        ref = ref + okw["bias"].reshape(1, -1, 1, 1)
    if "bn_relu" not in mode:
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


C3S_CASES = [c for c in C3_CASES if c[2] % 32 == 0]


@pytest.mark.parametrize("case", C3S_CASES, ids=["%dx%d->%d@%dx%d" % c for c in C3S_CASES])
@pytest.mark.parametrize("mode", ["online_u8_bn_relu_wino", "offline_s8_bias"])
def test_conv3x3_i8_sliced_vs_oracle(dev, ops, case, mode):
    """fq_weight_slices + fq_conv3x3_i8_sliced (BASELINE config 5: filters under Winograd-domain quantisation are not on one
    integer grid per channel): the three int8 digit slices and their power-of-two scale against the oracle, the convolution
    bit for bit against the oracle's exact integer form, and within 2^-20 of the fp64 convolution of the very tensors the
    reference's F.Convolution multiplies (fake-quantised activations x the given filter)."""
    n, cin, cout, h, w = case
    rng = np.random.default_rng(sum(case) + 19)
    x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
    signed = "s8" in mode
    if not signed:
        x = np.maximum(x, 0)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * rng.uniform(0.02, 1.0, (cout, 1, 1, 1))).astype(np.float32)
    if "wino" in mode:                                   # the filters config 5 really multiplies
        wt = O.wino_weight_fake_quant(wt, "F43", 8)[0]
    if cout >= 64:
        wt[1] = 0.0                                      # an all-zero filter
        wt[2, 0, 0, 0] = np.float32(2.0) ** -3           # its maximum a power of two
        wt[2] = np.clip(wt[2], -0.125, 0.125)
    codes, pscale, rowsum = ops.weight_slices_3x3(T(wt, dev))
    m, p = O.weight_slices(wt.transpose(0, 2, 3, 1).reshape(cout, -1))
    _eq(N(pscale), p, "per-channel power-of-two scale")
    rows_pad = (cout + 63) // 64 * 64
    for sl, d in enumerate(O.slice_digits(m)):
        got = N(codes)[sl, :rows_pad * 9 * cin].reshape(rows_pad, 9 * cin)[:cout]
        _eq(got, d.astype(np.int8), "digit slice %d" % sl)
        _eq(N(rowsum)[sl], d.sum(axis=1).astype(np.int32), "row sums of slice %d" % sl)
    kw, okw = {}, {}
    if mode.startswith("online"):
        stat = O.absmax_per_sample(x)
        kw.update(in_stat=T(stat, dev), width=8, flags=ops.act_flags(signed=signed))
        okw.update(in_max=O.batch_mean(stat), signed=signed, width=8)
    else:
        thr = np.float32(2.3)
        kw.update(in_thr=T(np.float32([thr]), dev), width=8, flags=ops.act_flags(signed=True))
        okw.update(in_max=thr, signed=True, width=8)
    if "bn_relu" in mode:
        sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
        sh = rng.standard_normal(cout).astype(np.float32)
        kw.update(bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu")
        okw.update(bn_scale=sc, bn_shift=sh, act="relu")
    if "bias" in mode:
        b = rng.standard_normal(cout).astype(np.float32)
        kw.update(bias=T(b, dev))
        okw.update(bias=b)
    cur = torch.zeros(1, device=dev)
    y, stat_out = ops.conv3x3_i8(T(x, dev), codes, pscale, rowsum, cur_out=cur, **kw)
    want = O.conv3x3_i8_sliced(x, wt, **okw)
    got = N(y)
    _eq(got, want, "sliced dense 3x3 convolution (exact integer sums of three digit slices)")
    _eq(N(stat_out), O.absmax_per_sample(got), "statistic")
    if "bn_relu" not in mode:
        # against the real thing: fp64 convolution of the fake-quantised activations with the filter as given
        xq = O.ste_forward(x, O.act_scale(okw["in_max"], okw["signed"], 8), okw["in_max"],
                           -okw["in_max"] if okw["signed"] else 0.0)
        ref = torch.nn.functional.conv2d(torch.from_numpy(xq.astype(np.float64)), torch.from_numpy(wt.astype(np.float64)),
                                         padding=1).numpy()
        if "bias" in mode:
            ref = ref + okw["bias"].reshape(1, -1, 1, 1)
        err = np.abs(got - ref).max()
        bound = 2.0 ** -20 * np.abs(wt).reshape(cout, -1).max(axis=1).max() * 9 * cin * np.abs(xq).max() + 1e-6 * np.abs(ref).max()
        assert err <= bound, (err, bound)
        np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6 * max(np.abs(ref).max(), 1e-6))


def test_division_by_double_reciprocal_is_ieee_exact(dev, ops):
    """fq_code divides with (float)((double)c * RN_f64(1/d)) (csrc: ieee_div_by).  Stress it where it could matter: inputs
    placed 0, +-1, +-2 ulp around every k + 0.5 rounding tie of the quotient, for many divisors, signed and unsigned."""
    rng = np.random.default_rng(123)
    for trial in range(40):
        width = int(rng.choice([8, 8, 4, 2, 7]))
        signed = bool(trial % 2)
        levels = (2 ** (width - 1) - 1) if signed else (2 ** width - 1)
        thr = np.float32(rng.uniform(1e-4, 300.0) if trial % 5 else rng.uniform(1e-12, 1e-6))
        scale = np.float32(thr / np.float32(levels))
        denom = np.float32(scale + np.float32(1e-10))
        ks = np.arange(-levels - 2 if signed else -2, levels + 3, dtype=np.float64)
        base = ((ks[:, None] + 0.5) * np.float64(denom)).astype(np.float32)            # quotient ~ k + 0.5
        xs = base.copy()
        for _ in range(3):
            xs = np.concatenate([np.nextafter(xs, np.float32(np.inf)), xs, np.nextafter(xs, np.float32(-np.inf))], axis=1)
        extra = (rng.standard_normal(4096) * float(thr)).astype(np.float32)
        x = np.concatenate([xs.reshape(-1), extra, np.float32([0.0, -0.0, thr, -thr, 1e-45, 3e38])]).astype(np.float32)
        x = np.resize(x, (4, (x.size + 3) // 4))
        want_y, _, _, want_codes = O.conv_input_fake_quant(x, signed, width, offline_threshold=thr)
        y, _, codes = ops.fake_quant_offline(T(x, dev), T(np.float32([thr]), dev), width, ops.act_flags(signed=signed),
                                             want_stat=False, want_codes=True)
        _eq(N(codes), want_codes.astype(np.int32), "codes (trial %d, thr %g, width %d)" % (trial, thr, width))
        _eq(N(y), want_y, "y")


# (the last three: the 3x3 form with its input rows staged in LDS on an odd height, one step per band, and bands of five steps with
# a last step of two rows - two steps of rows in flight)
@pytest.mark.parametrize("shape", [(2, 3, 32, 32), (3, 3, 33, 47), (1, 3, 224, 224), (5, 3, 18, 130), (3, 3, 45, 64), (70, 3, 50, 96),
                                   (260, 3, 76, 64)])
@pytest.mark.parametrize("mode", ["plain", "bn_relu", "bias_relu6"])
@pytest.mark.parametrize("ks,cout", [(3, 32), (7, 64)], ids=["3x3->32", "7x7->64"])
def test_stem_conv_s2_vs_oracle(dev, ops, shape, mode, ks, cout):
    """fq_stem_conv3x3s2 / fq_stem_conv7x7s2 (the un-quantised first convolution + BatchNorm + activation + statistic, on the
    fp32 matrix cores) against the oracle's fmaf-ordered restatement, BIT FOR BIT against the C++ twin's true fmaf chain
    and, loosely, against torch's convolution in fp64."""
    rng = np.random.default_rng(sum(shape) + len(mode) + ks)
    x = rng.standard_normal(shape).astype(np.float32)
    wt = (rng.standard_normal((cout, 3, ks, ks)) * 0.3).astype(np.float32)
    kw, okw = {}, {}
    if mode == "bn_relu":
        sc = rng.uniform(0.3, 1.5, cout).astype(np.float32)
        sh = rng.standard_normal(cout).astype(np.float32)
        kw.update(bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu")
        okw.update(bn_scale=sc, bn_shift=sh, act="relu")
    if mode == "bias_relu6":
        b = rng.standard_normal(cout).astype(np.float32)
        kw.update(bias=T(b, dev), act="relu6")
        okw.update(bias=b, act="relu6")
    y, stat = ops.stem_conv_s2(T(x, dev), T(wt, dev), **kw)
    want = O.stem_conv_s2(x, wt, **okw)
    got = N(y)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    assert (got != want).mean() < 1e-3                         # numpy's fmaf emulation differs only at double-rounding ties
    from oracle import host as H
    _eq(got, H.stem_conv_s2(x, wt, **okw), "the matrix-core accumulation IS the fmaf chain over (ci, ky, kx)")
    _eq(N(stat), O.absmax_per_sample(got), "statistic of the produced output")
    import torch.nn.functional as TF
    ref = TF.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None, stride=2, padding=ks // 2).numpy()
    if mode == "bn_relu":
        ref = np.maximum(ref * okw["bn_scale"].reshape(1, -1, 1, 1) + okw["bn_shift"].reshape(1, -1, 1, 1), 0)
    if mode == "bias_relu6":
        ref = np.clip(ref + okw["bias"].reshape(1, -1, 1, 1), 0, 6)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(2, 3, 224, 224), (3, 3, 200, 232), (1, 3, 195, 201), (5, 3, 31, 250), (130, 3, 64, 224),
                                   (40, 3, 96, 160), (128, 3, 224, 224), (6, 3, 75, 96), (200, 3, 75, 96)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("mode", ["bn_relu", "plain", "bias_relu6"])
def test_stem_conv7x7_with_maxpool_in_one_launch_vs_oracle(dev, ops, shape, mode):
    """fq_stem_conv7x7s2_pool (round 5): Conv2D(3 -> 64, 7x7, s2) -> BatchNorm -> activation -> MaxPool2D(3, 2, 1) with the
    convolution output kept in LDS, against `O.stem_conv_s2 -> O.bn_act_maxpool` (identity BatchNorm) and against the
    two-launch product path it replaces - bit for bit: odd output heights and widths (a last pooled row / column of two
    taps), bands of pooled rows with and without a halo row (batch 1 ... 130), signed values under the pool (no activation:
    the padding must not win)."""
    rng = np.random.default_rng(sum(shape) + len(mode))
    x = rng.standard_normal(shape).astype(np.float32)
    wt = (rng.standard_normal((64, 3, 7, 7)) * 0.3).astype(np.float32)
    kw, okw = {}, {}
    if mode == "bn_relu":
        sc = rng.uniform(0.3, 1.5, 64).astype(np.float32)
        sh = rng.standard_normal(64).astype(np.float32)
        kw.update(bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu")
        okw.update(bn_scale=sc, bn_shift=sh, act="relu")
    if mode == "bias_relu6":
        b = rng.standard_normal(64).astype(np.float32)
        kw.update(bias=T(b, dev), act="relu6")
        okw.update(bias=b, act="relu6")
    assert ops.stem_pool_supported(shape[2], shape[3])
    y, stat = ops.stem_conv_s2(T(x, dev), T(wt, dev), pool=True, **kw)
    # the product's two launches: the convolution, then pooling + statistic with the identity BatchNorm
    y2, _ = ops.stem_conv_s2(T(x, dev), T(wt, dev), **kw)
    one, zero = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    p2, stat2 = (ops.bn_act_maxpool_stat(y2, one, zero, "none", want_stat=True) if y2.shape[3] % 4 == 0 else (None, None))
    got = N(y)
    if p2 is not None:
        _eq(got, N(p2), "one launch == convolution launch + pooling launch")
        _eq(N(stat), N(stat2), "statistic of the pooled tensor")
    from oracle import host as H
    conv = H.stem_conv_s2(x, wt, **okw)
    want = O.bn_act_maxpool(conv, np.ones(64, np.float32), np.zeros(64, np.float32), "none")
    assert got.shape == want.shape
    _eq(got, want, "fmaf-chain convolution -> BatchNorm -> activation -> 3x3 / 2 max-pool")
    _eq(N(stat), O.absmax_per_sample(got), "statistic of the produced output")


def test_stem_pool_shapes_outside_the_built_range_are_refused(dev, ops):
    assert not ops.stem_pool_supported(600, 600) and not ops.stem_pool_supported(4, 4)
    x = torch.zeros(1, 3, 600, 600, device=dev)
    w = torch.zeros(64, 3, 7, 7, device=dev)
    with pytest.raises(ValueError, match="pooled form"):
        ops.stem_conv_s2(x, w, pool=True)


def test_stem_conv_rejects_other_shapes(dev, ops):
    from quantization.mxnet_amd._lib import FakeQuantError
    x = torch.zeros(1, 4, 8, 8, device=dev)
    w = torch.zeros(32, 4, 3, 3, device=dev)
    with pytest.raises(FakeQuantError):
        ops.stem_conv3x3s2(x, w)


@pytest.mark.parametrize("n,classes", [(128, 1000), (7, 10), (300, 37), (1, 1)])
def test_eval_counters_vs_oracle(dev, ops, n, classes):
    """fq_eval_counters (the CLI's argmax / per-class counters in one launch) against the oracle, with ties, NaN-free
    logits, out-of-range labels and accumulation over two batches."""
    rng = np.random.default_rng(n * 31 + classes)
    counters = torch.zeros(2 + 2 * classes, device=dev)
    want = None
    for batch in range(2):
        logits = rng.standard_normal((n, classes)).astype(np.float32)
        labels = rng.integers(0, classes, n).astype(np.int64)
        if classes > 3:
            logits[0, 1] = logits[0, 3] = np.float32(9.0)            # tie: the first index wins
            labels[0] = 1
            if n > 2:
                logits[1, 2] = logits[1, 0] = np.float32(9.0)
                labels[1] = 2                                         # tie lost: prediction is index 0
                labels[2] = classes + 5                               # outside: only `total` moves
        # make roughly half the predictions correct
        for i in range(0, n, 2):
            if 0 <= labels[i] < classes and i > 2:
                logits[i, labels[i]] = np.float32(20.0)
        ops.eval_counters(T(logits, dev), torch.from_numpy(labels).to(dev), counters)
        want = O.eval_counters(logits, labels, want)
    _eq(N(counters), want, "evaluation counters")
    assert N(counters)[1] == 2 * n


DENSE_EVAL_CASES = [(128, 1024, 1000), (7, 64, 10), (70, 100, 37), (1, 512, 3), (33, 2048, 1000)]


@pytest.mark.parametrize("n,cin,units", DENSE_EVAL_CASES, ids=["%dx%d->%d" % c for c in DENSE_EVAL_CASES])
@pytest.mark.parametrize("mode", ["online", "offline_channel_w4"])
def test_dense_i8_eval_vs_oracle(dev, ops, n, cin, units, mode):
    """fq_dense_i8_eval = the classifier on the codes + the evaluation counters of its logits in ONE launch: logits equal the
    oracle's integer form, counters equal the oracle's counters of those logits, accumulated over three calls that reuse the
    library-side workspace (it must come back zeroed).  Ties (two identical weight rows: the first index wins, whichever
    workgroup finishes last), a NaN logit (a NaN bias: the maximum), out-of-range labels, partial sample and unit tiles."""
    from oracle import host as H
    rng = np.random.default_rng(n + cin * 7 + units)
    per_channel = "channel" in mode
    wt_width = 4 if "w4" in mode else 8
    wt = (rng.standard_normal((units, cin)) * rng.uniform(0.05, 1.0, (units, 1))).astype(np.float32)
    if units > 40:
        wt[5] = np.abs(wt[5]) + 1                                     # (inputs are >= 0: unit 5 is every sample's maximum ...)
        wt[37] = wt[5]                                                # ... with equal logits in another unit tile
        wt[6] = wt[5]                                                 # ... and inside its own
    b = rng.standard_normal(units).astype(np.float32)
    if units > 40:
        b[37] = b[6] = b[5] = np.float32(5)
    codes, scales, rowsum = ops.weight_codes(T(wt, dev), 1 if per_channel else units, wt_width)
    counters = torch.zeros(2 + 2 * units, device=dev)
    want_c = None
    for call in range(3):
        x = np.maximum(rng.standard_normal((n, cin)) * 2, 0).astype(np.float32)
        bias = b.copy()
        if call == 2 and units > 2:
            bias[2] = np.nan                                          # every prediction of this call is unit 2
        labels = rng.integers(0, units, n).astype(np.int64)
        if n > 3:
            labels[1] = units + 3                                     # only `total` moves
            labels[2] = 5 if units > 5 else 0
            x[2] = 0
            if units > 40:
                labels[3] = 5                                         # units 5 / 6 / 37 tie for the maximum: 5 wins
        kw, okw = {}, {}
        stat = O.absmax_per_sample(x)
        flags = ops.act_flags(signed=False, lo_neg_max=False)
        if mode.startswith("online"):
            kw.update(in_stat=T(stat, dev), width=8, flags=flags)
            okw.update(in_max=O.batch_mean(stat), signed=False, width=8)
        else:
            thr = np.float32(3.1)
            kw.update(in_thr=T(np.float32([thr]), dev), in_stat=T(stat, dev), width=8, flags=flags)
            okw.update(in_max=thr, signed=False, width=8)
        cur = torch.zeros(1, device=dev)
        y = ops.dense_i8_eval(T(x, dev), codes, scales, rowsum, torch.from_numpy(labels).to(dev), counters, bias=T(bias, dev),
                              cur_out=cur, **kw)
        want = O.pwconv_i8(x.reshape(n, cin, 1, 1), wt.reshape(units, cin, 1, 1), 1 if per_channel else units, wt_width,
                           bias=bias, **okw).reshape(n, units)
        _eq(N(y), want, "logits of call %d" % call)
        _eq(N(cur), np.float32([O.batch_mean(stat)]), "current_input_max")
        want_c = O.eval_counters(want, labels, want_c)
        _eq(N(counters), want_c, "counters after call %d" % call)
        # the two-launch formulation and the host twin give the same
        y2, _ = ops.pwconv_i8(T(x, dev).reshape(n, cin, 1, 1), codes, scales, rowsum, bias=T(bias, dev), want_stat=False, **kw)
        _eq(N(y2).reshape(n, units), want, "fq_pwconv_i8 on planes of one pixel")
        if call == 0:
            hy, hc = H.dense_i8_eval(x, wt, 1 if per_channel else units, wt_width, labels, None, bias=bias, **okw)
            _eq(hy, want, "host twin: logits")
            _eq(hc, O.eval_counters(want, labels, None), "host twin: counters")
    assert N(counters)[1] == 3 * n
    if units > 40 and n > 3:
        assert N(counters)[2 + 5] >= 2                                # the tie went to the first index (calls 0 and 1)


@pytest.mark.parametrize("n,l,k,cout,zoff", [(2, 49, 27, 32, 128), (3, 100, 288, 70, 0), (1, 7, 9, 5, 128),
                                              (4, 196, 1152, 96, 128), (1, 33, 64, 33, 0)])
def test_gemm_i8_codes_is_exact(dev, ops, n, l, k, cout, zoff):
    """fq_gemm_i8_codes: int8 x int8 -> int32 on the matrix cores, bit-exact against integer matmul, including
    accumulators far beyond 2^24 (the fp32 formulation's limit), ragged K / Cout / column counts and the 128 re-centring."""
    rng = np.random.default_rng(n * 1000 + l + k + cout)
    if zoff:
        xu = rng.integers(0, 256, (n * l, k))                          # unsigned codes ...
        xs = (xu - 128).astype(np.int8)                                # ... as stored
    else:
        xu = rng.integers(-127, 128, (n * l, k))
        xs = xu.astype(np.int8)
    if k >= 1000:
        xu[:, :] = 255 if zoff else 127                                # drive |acc| past 2^24: 1152 * 255 * 127 = 3.7e7
        xs = (xu - zoff).astype(np.int8)
    w = rng.integers(-127, 128, (cout, k)).astype(np.int8)
    if k >= 1000:
        w[::2, :] = 127
    got = ops.gemm_i8_codes(torch.from_numpy(xs).to(dev), torch.from_numpy(w).to(dev), n, l, zoff)
    want = (xu.astype(np.int64) @ w.astype(np.int64).T).reshape(n, l, cout).transpose(0, 2, 1)
    assert got.dtype == torch.int32 and tuple(got.shape) == (n, cout, l)
    _eq(N(got).astype(np.int64), want, "integer GEMM")
    _eq(N(got), O.gemm_i8_codes(xs, w, n, l, zoff), "oracle")
    if k >= 1000:
        assert np.abs(want).max() > 2 ** 24


@pytest.mark.parametrize("shape", [(128, 1024, 7, 7), (3, 17, 5, 4), (2, 256, 1, 1), (5, 512, 14, 14)])
def test_global_avg_pool_stat_vs_oracle(dev, ops, shape):
    rng = np.random.default_rng(sum(shape))
    x = np.maximum(rng.standard_normal(shape), 0).astype(np.float32) * 3
    y, stat = ops.global_avg_pool_stat(T(x, dev))
    want = O.global_avg_pool(x)
    _eq(N(y), want, "pooled")
    _eq(N(stat), O.absmax_per_sample(want), "statistic")
    np.testing.assert_allclose(N(y), x.mean(axis=(2, 3), keepdims=True), rtol=2e-6, atol=1e-7)


# ---- fused producers at the benchmark's full batch (BASELINE configs[1]: mobilenet1.0, batch 128) ----------------------
def _subset_pins_full_batch(run, x_full, picks, oracle_fn, what, exact=True):
    """`run(x)` -> (y, stat).  In offline mode a sample's output cannot depend on its batch-mates: the full-batch result
    on `picks` must be BIT-IDENTICAL to a small-batch call (different grid / tile boundaries), and that small call is
    checked against the oracle - which pins the full-size kernel to the oracle on those samples.  The per-sample
    statistic must be the abs-max of what was actually written, for every sample."""
    y_full, stat_full = run(x_full)
    xs = x_full[picks].contiguous()
    y_small, stat_small = run(xs)
    assert torch.equal(y_full[picks], y_small), what + ": full-batch output differs from the small-batch call"
    assert torch.equal(stat_full[picks], stat_small), what + ": statistic differs"
    assert torch.equal(stat_full, y_full.abs().reshape(y_full.shape[0], -1).amax(dim=1)), what + ": statistic != max|y|"
    want = oracle_fn(xs.cpu().numpy())
    got = y_small.cpu().numpy()
    if exact:
        _eq(got, want, what + " vs oracle")
    else:
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6, err_msg=what)
        assert (got != want).mean() < 1e-3


def test_fused_producers_full_batch_properties(dev, ops):
    """stem (128,3,224,224); depthwise 32@112x112 stride 1 and 64@112x112 stride 2; pointwise 32->64@112x112 (stream
    form), 512->512@14x14 (chunked form), 512->1024@7x7 and 1024->1024@7x7 (tile form) - the largest instance of every
    producer kernel family in the benchmark step."""
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    rng = np.random.default_rng(5)
    picks = torch.tensor([0, 63, 127], device=dev)
    thr = np.float32(2.1)
    thr_t = T(np.float32([thr]), dev)

    def bn(c):
        return rng.uniform(0.4, 1.4, c).astype(np.float32), rng.standard_normal(c).astype(np.float32)

    # stem
    x = torch.randn(128, 3, 224, 224, device=dev, generator=g)
    w = (rng.standard_normal((32, 3, 3, 3)) * 0.3).astype(np.float32)
    sc, sh = bn(32)
    _subset_pins_full_batch(lambda t: ops.stem_conv3x3s2(t, T(w, dev), bn_scale=T(sc, dev), bn_shift=T(sh, dev), act="relu"),
                            x, picks, lambda a: O.stem_conv3x3s2(a, w, None, sc, sh, "relu"), "stem", exact=False)
    del x
    # depthwise
    for c, hw, s in [(32, 112, 1), (64, 112, 2), (512, 14, 1)]:
        x = torch.relu(torch.randn(128, c, hw, hw, device=dev, generator=g)) * 1.5
        w = (rng.standard_normal((c, 1, 3, 3)) * 0.4).astype(np.float32)
        sc, sh = bn(c)
        _subset_pins_full_batch(
            lambda t: ops.dwconv3x3(t, T(w, dev), None, stride=s, in_thr=thr_t, width=8, flags=0, bn_scale=T(sc, dev),
                                    bn_shift=T(sh, dev), act="relu"),
            x, picks, lambda a: O.dwconv3x3(a, w, None, s, thr, False, 8, None, sc, sh, "relu"),
            "depthwise %d@%d s%d" % (c, hw, s), exact=False)
        del x
    # pointwise: stream / chunk / tile forms
    for cin, cout, hw in [(32, 64, 112), (512, 512, 14), (512, 1024, 7), (1024, 1024, 7)]:
        x = torch.relu(torch.randn(128, cin, hw, hw, device=dev, generator=g)) * 1.5
        w = (rng.standard_normal((cout, cin, 1, 1)) * 0.1).astype(np.float32)
        sc, sh = bn(cout)
        codes, scales, rowsum = ops.weight_codes(T(w, dev), cout, 8)
        _subset_pins_full_batch(
            lambda t: ops.pwconv_i8(t, codes, scales, rowsum, in_thr=thr_t, width=8, flags=0, bn_scale=T(sc, dev),
                                    bn_shift=T(sh, dev), act="relu"),
            x, picks, lambda a: O.pwconv_i8(a, w, cout, 8, in_max=thr, signed=False, width=8, bn_scale=sc, bn_shift=sh,
                                            act="relu"),
            "pointwise %d->%d@%d" % (cin, cout, hw))
        # exact linearity in a power-of-two weight scale (codes identical, scales doubled), without BN / activation
        y1, _ = ops.pwconv_i8(x, codes, scales, rowsum, in_thr=thr_t, width=8, flags=0)
        c2, s2, r2 = ops.weight_codes(T(w * 2, dev), cout, 8)
        assert torch.equal(c2, codes)
        y2, _ = ops.pwconv_i8(x, c2, s2, r2, in_thr=thr_t, width=8, flags=0)
        assert torch.equal(y2, y1 * 2), "pointwise %d->%d: not linear in the weight scale" % (cin, cout)
        del x, y1, y2
