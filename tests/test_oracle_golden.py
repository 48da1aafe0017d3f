"""The oracle (oracle/fq_oracle.py) against the golden vectors produced by the reference's own Python
(tools/gen_golden.py).  CPU only.  Bar: integer stages bit-exact, dequantised floats bit-exact where the op order is
defined, 1e-6 relative otherwise (stated per test)."""
import numpy as np
import pytest

from oracle import fq_oracle as O


def _eq(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), "%s: %d mismatches, max |d|=%g" % (
        what, int((a != b).sum()), float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))))


# ---- G1-G3: pinned against the real reference module -----------------------------------------------------
@pytest.mark.parametrize("name", ["halfnormal", "exponential", "relu_outlier", "tiny_range", "shape4d"])
def test_histogram_matches_reference(golden, name):
    g = golden("g1_histogram")
    fm = g[name + "/fm"]
    h, m = O.discrete_histogram(fm, 2048, None)
    _eq(h, g[name + "/hist_auto"], "auto hist")
    assert np.float32(m) == g[name + "/max_auto"]
    h2, _ = O.discrete_histogram(fm, 2048, g[name + "/max_fixed"])
    _eq(h2, g[name + "/hist_fixed"], "fixed-range hist (clip into last bin)")
    h3, _ = O.discrete_histogram(fm, 128, None)
    _eq(h3, g[name + "/hist_auto_b128"], "128-bin hist")
    assert h.sum() == np.count_nonzero(fm)          # zeros are dropped (distribution_calibrate.py:40)


@pytest.mark.parametrize("name", ["halfnormal", "sparse"])
@pytest.mark.parametrize("levels", [16, 32])
def test_kl_small_bins_matches_reference(golden, name, levels):
    g = golden("g2_kl")
    assert O.kl_calibrate(g[name + "/hist_b256"], levels, levels, 256) == int(g["%s/best_b256_L%d" % (name, levels)])


@pytest.mark.parametrize("name,levels", [("halfnormal", 256), ("exponential", 128), ("relu_outlier", 16),
                                         ("accumulated6", 256), ("sparse", 8), ("spike", 128)])
def test_kl_2048_matches_reference(golden, name, levels):
    g = golden("g2_kl")
    best = O.kl_calibrate(g[name + "/hist"], levels, levels, 2048)
    assert best == int(g["%s/best_L%d" % (name, levels)])
    assert O.kl_threshold(best, g[name + "/fm_max"], 2048) == g["%s/thr_L%d" % (name, levels)]


def test_collect_feature_maps_matches_reference(golden):
    g = golden("g3_collect")
    batches = g["batches"]
    for k in range(3):
        hist, fm_max = None, None
        for b in batches:
            fm = np.maximum(b, 0) * np.float32(k + 1)
            if k == 2:
                fm = fm[:, :, ::2, ::2]
            h, m = O.discrete_histogram(fm, 2048, fm_max)
            if fm_max is None:
                fm_max = m                       # first batch fixes the range (distribution_calibrate.py:97-101)
            hist = h if hist is None else hist + h
        _eq(hist, g["hist%d" % k], "accumulated hist block %d" % k)
        assert fm_max == g["fm_max%d" % k]


# ---- G4: activation branch ---------------------------------------------------------------------------------
def _act_cases(g, prefix):
    return sorted({k.split("/")[0] for k in g if k.startswith(prefix)})


def test_conv_activation_fake_quant(golden):
    g = golden("g4_activation")
    tags = [t for t in _act_cases(g, "conv_") if t != "conv_zero"]
    assert len(tags) == 12
    for tag in tags:
        signed = "_s_" in tag
        width = int(tag.rsplit("w", 1)[1])
        x = g[tag + "/x"]
        y, cur, scale, codes = O.conv_input_fake_quant(x, signed, width)
        assert cur == g[tag + "/online_max"], tag
        assert scale == g[tag + "/online_scale"], tag
        _eq(codes, g[tag + "/online_codes"], tag + " online codes")
        _eq(y, g[tag + "/online_y"], tag + " online y")
        lim = 2 ** (width - 1) - 1 if signed else 2 ** width - 1
        assert codes.max() <= lim and codes.min() >= (-lim if signed else 0)
        y, cur2, scale, codes = O.conv_input_fake_quant(x, signed, width, offline_threshold=g[tag + "/offline_thr"])
        assert cur2 == g[tag + "/offline_curmax"]         # statistic still computed in offline mode (:56)
        assert scale == g[tag + "/offline_scale"], tag
        _eq(codes, g[tag + "/offline_codes"], tag + " offline codes")
        _eq(y, g[tag + "/offline_y"], tag + " offline y")


def test_conv_activation_all_zero(golden):
    g = golden("g4_activation")
    y, cur, scale, _ = O.conv_input_fake_quant(g["conv_zero/x"])
    assert cur == 0 and scale == 0
    _eq(y, g["conv_zero/online_y"])


def test_dense_activation_fake_quant(golden):
    g = golden("g4_activation")
    for tag in _act_cases(g, "dense_"):
        signed = "_s_" in tag
        width = int(tag.rsplit("w", 1)[1])
        x = g[tag + "/x"]
        y, cur, _, _ = O.dense_input_fake_quant(x, signed, width)
        assert cur == g[tag + "/online_max"]
        _eq(y, g[tag + "/online_y"], tag)
        assert y.min() >= 0                                # clip_min defaults to 0 even when signed
        y, _, _, _ = O.dense_input_fake_quant(x, signed, width, offline_threshold=g[tag + "/offline_thr"])
        _eq(y, g[tag + "/offline_y"], tag + " offline")


def test_dense_on_an_unflattened_input_golden(golden):
    """G12: the reference's `_dense_forward` on (N, C, H, W) inputs (vgg's first Dense): `F.max(F.abs(x), axis=1).mean()` reduces
    over C only.  The oracle follows the line - and differs from the per-sample statistic a flattened input gives."""
    g = golden("g12_dense_unflattened")
    tags = sorted({k.split("/")[0] for k in g})
    assert len(tags) == 6
    for tag in tags:
        signed = tag.endswith("_s")
        x = g[tag + "/x"]
        y, cur, _, _ = O.dense_input_fake_quant(x, signed, 8)
        assert cur == g[tag + "/online_max"], tag
        _eq(y, g[tag + "/online_y"], tag)
        if x.shape[2] * x.shape[3] > 1:
            assert cur != O.dense_input_fake_quant(x.reshape(x.shape[0], -1), signed, 8)[1]
        y, cur_off, _, _ = O.dense_input_fake_quant(x, signed, 8, offline_threshold=g[tag + "/offline_thr"])
        _eq(y, g[tag + "/offline_y"], tag + " offline")
        assert cur_off == g[tag + "/offline_curmax"]


# ---- G5/G6: weights ------------------------------------------------------------------------------------------
def test_weight_fake_quant(golden):
    g = golden("g5_weight")
    groups = {"dw16": 16, "pw32x16": 1, "c8x4k3": 1}
    n = 0
    for name, grp in groups.items():
        w = g[name + "/w"]
        for qt in ("layer", "group", "channel"):
            for width in (8, 4):
                key = "%s/%s_w%d" % (name, qt, width)
                if key not in g:
                    continue
                wq, _ = O.weight_fake_quant(w, qt, width, num_group=grp)
                _eq(wq, g[key], key)
                n += 1
    assert n == 18
    for qt in ("layer", "channel"):
        for width in (8, 4):
            wq, _ = O.weight_fake_quant(g["dense/%s_w%d/w" % (qt, width)], qt, width)
            _eq(wq, g["dense/%s_w%d/wq" % (qt, width)], "dense " + qt)


@pytest.mark.parametrize("variant", ["F23", "F43", "F63"])
def test_winograd_weight_fake_quant(golden, variant):
    g = golden("g6_winograd")
    _eq(O.winograd_G(variant), g[variant + "/G"], "G")
    for name in ("c8x4k3", "dw16"):
        for width in (8, 4):
            w = g["%s/%s_w%d/w" % (variant, name, width)]
            wq, _, _ = O.wino_weight_fake_quant(w, variant, width, GI=g[variant + "/GI"], GTI=g[variant + "/GTI"])
            _eq(wq, g["%s/%s_w%d/wq" % (variant, name, width)], "%s %s w%d" % (variant, name, width))


# ---- G7: EMA --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["layer_w8", "channel_w4"])
def test_ema_sequence(golden, tag):
    g = golden("g7_g9_ema_state")
    cur, ema = g[tag + "/calib_cur"], g[tag + "/calib_ema"]
    state = np.zeros(cur.shape[1], np.float32)              # input_max starts at 0 (initialize.py:72-73)
    for step in range(cur.shape[0]):
        state = O.ema_update(state, cur[step], 0.9)
        _eq(state, ema[step], "ema step %d" % step)


# ---- G8: int-code path ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["u01", "normal", "shifted"])
@pytest.mark.parametrize("t", ["int8", "uint8"])
def test_quantize_codes(golden, name, t):
    g = golden("g8_quantized_conv")
    codes, scale = O.quantize_codes(g[name + "/x"], t)
    assert codes.dtype == np.int32
    _eq(codes, g["%s/%s_codes" % (name, t)])
    assert scale == g["%s/%s_scale" % (name, t)]
    _eq(O.dequantize(codes, scale), g["%s/%s_deq" % (name, t)])


@pytest.mark.parametrize("use_bias", [0, 1])
@pytest.mark.parametrize("groups", [1, 2])
def test_quantized_conv_three_way(golden, use_bias, groups):
    """The shape of the reference's tests/test_quantized_conv.py:36-57, asserted: int-code conv == golden; and
    int-code conv ~ simulated conv ~ float conv within quantisation error."""
    g = golden("g8_quantized_conv")
    tag = "conv_b%d_g%d" % (use_bias, groups)
    x, w, b = g[tag + "/x"], g[tag + "/w"], (g[tag + "/b"] if use_bias else None)
    y = O.qconv2d_forward(x, w, b, (1, 1), (1, 1), groups)
    np.testing.assert_allclose(y, g[tag + "/y_int"], rtol=0, atol=0)
    yf = O.qconv2d_forward(x, w, b, (1, 1), (1, 1), groups, quantized=False)
    np.testing.assert_allclose(yf, g[tag + "/y_float"], rtol=1e-5, atol=1e-5)
    assert np.abs(y - g[tag + "/y_sim"]).max() < 0.1 and np.abs(y - yf).max() < 0.1


def test_roundf_ties_away_from_zero():
    x = np.float32([-2.5, -1.5, -0.5, 0.5, 1.5, 2.5, 0.49999997, -0.49999997, 8388609.0, -0.0])
    _eq(O.roundf(x), np.float32([-3, -2, -1, 1, 2, 3, 0, -0.0, 8388609.0, -0.0]))
