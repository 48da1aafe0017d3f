"""The C++/OpenMP restatement (oracle/fq_host.cpp -> oracle/libfq_host.so, include/fakequant_host.h) pinned bit-for-bit
against the golden vectors made from the reference's own Python and against the numpy oracle on seeded inputs.  CPU only.
Once pinned here it serves as the fast oracle for the full-size GPU parity tests and as bench.py's CPU baseline."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import fq_oracle as O
from oracle import host as H


def _eq(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), "%s: %d mismatches, max |d|=%g" % (
        what, int((a != b).sum()), float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))))


def test_host_library_exports_every_declared_symbol():
    import ctypes
    text = open(os.path.join(ROOT, "include", "fakequant_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(fq_[a-z0-9_]+_host)\s*\(", text)))
    assert len(declared) >= 30
    lib = ctypes.CDLL(H.build())
    for name in declared:
        assert hasattr(lib, name), name
    # every device entry point that computes has a host twin
    dev = open(os.path.join(ROOT, "include", "fakequant.h")).read()
    dev = re.sub(r"/\*.*?\*/", "", dev, flags=re.S)
    skip = {"fq_device_info", "fq_profile_enable", "fq_profile_reset", "fq_profile_read", "fq_profile_calibrate",
            "fq_act_workspace_bytes", "fq_pwconv_workspace_bytes", "fq_weight_workspace_bytes",
            "fq_kl_workspace_bytes", "fq_dense_i8_eval_workspace_bytes", "fq_build_id", "fq_profile_launch_overhead",
            "fq_qconv_weights_bytes", "fq_qconv_kind", "fq_qconv_workspace_bytes", "fq_qconv_workspace_init",
            "fq_profile_read_moved", "fq_stem_conv7x7s2_pool_supported", "fq_pwconv_i8_stat_supported",
            "fq_pwdw_fused_supported", "fq_build_has", "fq_debug_fast_quotient", "fq_pwconv_i8_sub2_supported", "fq_pwconv_i8_gap_supported", "fq_pwconv_i8_shortcut_supported",
            # transport, not arithmetic: the RCCL collectives of multi-GPU calibration
            "fq_comm_unique_id", "fq_comm_init", "fq_comm_world", "fq_allreduce_f32", "fq_allreduce_f64",
            "fq_allreduce_i64", "fq_comm_destroy"}
    for name in sorted(set(re.findall(r"\b(fq_[a-z0-9_]+)\s*\(", dev)) - skip):
        assert name + "_host" in declared, "no host twin for " + name


# ---- against the golden vectors (the reference's own code) ---------------------------------------------------------
@pytest.mark.parametrize("name", ["halfnormal", "exponential", "relu_outlier", "tiny_range", "shape4d"])
def test_histogram_golden(golden, name):
    g = golden("g1_histogram")
    fm = g[name + "/fm"]
    h, m = H.discrete_histogram(fm, 2048, None)
    _eq(h, g[name + "/hist_auto"])
    assert m == g[name + "/max_auto"]
    _eq(H.discrete_histogram(fm, 2048, g[name + "/max_fixed"])[0], g[name + "/hist_fixed"])
    _eq(H.discrete_histogram(fm, 128, None)[0], g[name + "/hist_auto_b128"])


def test_kl_golden(golden):
    g = golden("g2_kl")
    for name, levels in [("halfnormal", 256), ("exponential", 128), ("relu_outlier", 16), ("accumulated6", 256),
                         ("sparse", 8), ("spike", 128)]:
        assert int(H.kl_search(g[name + "/hist"], levels, levels)[0]) == int(g["%s/best_L%d" % (name, levels)])
    for name in ("halfnormal", "sparse"):
        for levels in (16, 32):
            assert int(H.kl_search(g[name + "/hist_b256"], levels, levels)[0]) == \
                int(g["%s/best_b256_L%d" % (name, levels)])
    # several layers in one call
    hs = np.stack([g[n + "/hist"] for n in ("halfnormal", "accumulated6")])
    _eq(H.kl_search(hs, 256, 256), np.int32([g["halfnormal/best_L256"], g["accumulated6/best_L256"]]))


def test_activation_golden(golden):
    g = golden("g4_activation")
    tags = sorted({k.split("/")[0] for k in g if k.startswith("conv_")})
    for tag in tags:
        x = g[tag + "/x"]
        if tag == "conv_zero":
            _eq(H.fake_quant_online(x)[0], g["conv_zero/online_y"])
            continue
        signed = "_s_" in tag
        width = int(tag.rsplit("w", 1)[1])
        fl = H.act_flags(signed)
        y, cur, codes = H.fake_quant_online(x, width, fl, want_codes=True)
        assert cur == g[tag + "/online_max"]
        _eq(codes, g[tag + "/online_codes"].astype(np.int32), tag)
        _eq(y, g[tag + "/online_y"], tag)
        y, cur, codes = H.fake_quant_offline(x, g[tag + "/offline_thr"], width, fl, want_codes=True)
        assert cur == g[tag + "/offline_curmax"]
        _eq(codes, g[tag + "/offline_codes"].astype(np.int32), tag)
        _eq(y, g[tag + "/offline_y"], tag)
        _eq(H.unfused_chain(x, width, fl)[0], g[tag + "/online_y"], tag + " unfused chain")
    for tag in sorted({k.split("/")[0] for k in g if k.startswith("dense_")}):
        signed = "_s_" in tag
        width = int(tag.rsplit("w", 1)[1])
        fl = H.act_flags(signed, lo_neg_max=False)
        _eq(H.fake_quant_online(g[tag + "/x"], width, fl)[0], g[tag + "/online_y"], tag)
        _eq(H.fake_quant_offline(g[tag + "/x"], g[tag + "/offline_thr"], width, fl)[0], g[tag + "/offline_y"], tag)


def test_weight_golden(golden):
    g = golden("g5_weight")
    for name, grp in {"dw16": 16, "pw32x16": 1, "c8x4k3": 1}.items():
        w = g[name + "/w"]
        for qt in ("layer", "group", "channel"):
            for width in (8, 4):
                key = "%s/%s_w%d" % (name, qt, width)
                if key in g:
                    rows = {"layer": 1, "group": grp, "channel": w.shape[0]}[qt]
                    _eq(H.weight_fake_quant(w, rows, width)[0], g[key], key)
    for qt in ("layer", "channel"):
        for width in (8, 4):
            w = g["dense/%s_w%d/w" % (qt, width)]
            _eq(H.weight_fake_quant(w, 1 if qt == "layer" else w.shape[0], width)[0], g["dense/%s_w%d/wq" % (qt, width)])


@pytest.mark.parametrize("variant", ["F23", "F43", "F63"])
def test_winograd_golden(golden, variant):
    g = golden("g6_winograd")
    for name in ("c8x4k3", "dw16"):
        for width in (8, 4):
            w = g["%s/%s_w%d/w" % (variant, name, width)]
            wq, _ = H.wino_weight_fake_quant(w, g[variant + "/G"], g[variant + "/GI"], g[variant + "/GTI"], width)
            _eq(wq, g["%s/%s_w%d/wq" % (variant, name, width)])


def test_ema_and_codes_golden(golden):
    g = golden("g7_g9_ema_state")
    for tag in ("layer_w8", "channel_w4"):
        cur, ema = g[tag + "/calib_cur"], g[tag + "/calib_ema"]
        state = np.zeros(cur.shape[1], np.float32)
        for step in range(cur.shape[0]):
            state = H.ema_update(state, cur[step], 0.9)
            _eq(state, ema[step])
    g = golden("g8_quantized_conv")
    for name in ("u01", "normal", "shifted"):
        for t in ("int8", "uint8"):
            codes, rng = H.quantize_codes(g[name + "/x"], t)
            _eq(codes, g["%s/%s_codes" % (name, t)])
            assert rng[2] == g["%s/%s_scale" % (name, t)]
            _eq(H.dequantize(codes, rng[2]), g["%s/%s_deq" % (name, t)])


# ---- against the numpy oracle on seeded inputs (ragged shapes, ties, negatives, every mode) ---------------------------
SHAPES = [(4, 8, 7, 7), (3, 5, 9, 11), (2, 32, 14, 14), (5, 1000), (1, 3, 17, 5)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("signed,width", [(False, 8), (True, 8), (False, 4), (True, 4), (False, 2)])
def test_activation_vs_numpy_oracle(shape, signed, width):
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(shape) * 2.5).astype(np.float32)
    x.reshape(-1)[::7] = 0
    want_y, want_cur, scale, want_codes = O.conv_input_fake_quant(x, signed, width)
    # exact ties of the quotient: multiples of half the scale
    x.reshape(-1)[1::11] = (np.arange(x.reshape(-1)[1::11].size, dtype=np.float32) + np.float32(0.5)) * scale
    want_y, want_cur, scale, want_codes = O.conv_input_fake_quant(x, signed, width)
    y, cur, codes = H.fake_quant_online(x, width, H.act_flags(signed), want_codes=True)
    assert cur == want_cur
    _eq(codes, want_codes.astype(np.int32))
    _eq(y, want_y)
    _eq(H.absmax_per_sample(x), O.absmax_per_sample(x))
    y2, _, _ = H.fake_quant_online_prestat(x, O.absmax_per_sample(x), width, H.act_flags(signed))
    _eq(y2, want_y)
    thr = np.float32(1.7)
    _eq(H.fake_quant_offline(x, thr, width, H.act_flags(signed))[0], O.conv_input_fake_quant(x, signed, width, thr)[0])
    # (the Dense block's clip range on its FLATTENED input; an un-flattened one has its own statistic, convert_dense.py:41)
    _eq(H.fake_quant_online(x, width, H.act_flags(signed, lo_neg_max=False))[0].reshape(x.shape[0], -1),
        O.dense_input_fake_quant(x.reshape(x.shape[0], -1), signed, width)[0])
    a = np.abs(x) + np.float32(0.1)
    ya, cura, _ = H.fake_quant_online(a, width, H.act_flags(no_abs=True, no_eps=True))
    wa, wcur, _, _ = O.act_output_fake_quant(a, width)
    assert cura == wcur
    _eq(ya, wa)


def test_fused_producers_vs_numpy_oracle():
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((3, 8, 9, 10)) * 2).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 8).astype(np.float32)
    sh = rng.standard_normal(8).astype(np.float32)
    for act in ("relu", "relu6", "none"):
        y, stat = H.bn_act(x, sc, sh, act, want_stat=True)
        _eq(y, O.bn_act(x, sc, sh, act))
        _eq(stat, O.absmax_per_sample(y))
    _eq(H.global_avg_pool(x), O.global_avg_pool(x))
    w = (rng.standard_normal((8, 1, 3, 3)) * 0.3).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    for stride in (1, 2):
        for in_max in (None, np.float32(2.2)):
            for signed in (False, True):
                want = O.dwconv3x3(x, w, b, stride, in_max, signed, 8, None, sc, sh, "relu6")
                _eq(H.dwconv3x3(x, w, b, stride, in_max, signed=signed, bn_scale=sc, bn_shift=sh, act="relu6"), want)
    stat = O.absmax_per_sample(x)
    _eq(H.dwconv3x3(x, w, None, 1, in_stat=stat, act="relu"), O.dwconv3x3(x, w, None, 1, O.batch_mean(stat), act="relu"))
    xs = rng.standard_normal((2, 3, 12, 14)).astype(np.float32)
    ws = (rng.standard_normal((32, 3, 3, 3)) * 0.2).astype(np.float32)
    sc32, sh32 = rng.uniform(0.5, 1.5, 32).astype(np.float32), rng.standard_normal(32).astype(np.float32)
    _eq(H.stem_conv3x3s2(xs, ws, None, sc32, sh32, "relu"), O.stem_conv3x3s2(xs, ws, None, sc32, sh32, "relu"))
    # pointwise on integer codes: per-layer and per-channel weights, 8 and 4 bit, signed and unsigned inputs
    wp = (rng.standard_normal((24, 8, 1, 1)) * 0.2).astype(np.float32)
    for rps, ww in ((24, 8), (1, 8), (1, 4)):
        for signed in (False, True):
            want = O.pwconv_i8(x, wp, rps, ww, np.float32(2.0), signed, 8, None, None, sc[:1].repeat(24), sh[:1].repeat(24),
                               "relu")
            got = H.pwconv_i8(x, wp, rps, ww, np.float32(2.0), signed=signed, bn_scale=sc[:1].repeat(24),
                              bn_shift=sh[:1].repeat(24), act="relu")
            _eq(got, want)
    codes, scales, rowsum = H.weight_codes(wp, 1, 4)
    wc, ws_ = O.weight_codes(wp, 1, 4)
    _eq(codes[:24, :8].astype(np.int32), wc)
    _eq(scales, ws_)
    _eq(rowsum, wc.sum(axis=1).astype(np.int32))


def test_round2_consumers_vs_numpy_oracle():
    """The entry points added in round 2 - dense 3x3 on the integer codes, strided 1x1, the residual operand, the offline
    side statistic, BatchNorm + activation + max pooling, the 7x7 first convolution: C++ twin against the numpy
    restatement, and the numpy restatement against an independent formulation where there is one."""
    import torch
    rng = np.random.default_rng(21)
    x = np.maximum(rng.standard_normal((3, 64, 7, 9)) * 2, 0).astype(np.float32)
    stat = O.absmax_per_sample(x)
    in_max = O.batch_mean(stat)
    sc = rng.uniform(0.5, 1.5, 96).astype(np.float32)
    sh = rng.standard_normal(96).astype(np.float32)
    # dense 3x3
    w3 = (rng.standard_normal((96, 64, 3, 3)) * 0.2).astype(np.float32)
    for rps, ww, signed in ((96, 8, False), (1, 8, True), (1, 4, False)):
        want = O.conv3x3_i8(x, w3, rps, ww, in_max, signed, 8, None, None, sc, sh, "relu")
        _eq(H.conv3x3_i8(x, w3, rps, ww, in_max=in_max, signed=signed, bn_scale=sc, bn_shift=sh, act="relu"), want)
    plain = O.conv3x3_i8(x, w3, 1, 8, in_max)
    wq = O.weight_fake_quant(w3, "channel", 8)[0]
    xq = O.ste_forward(x, O.act_scale(in_max, False, 8), in_max, 0.0)
    ref = torch.nn.functional.conv2d(torch.from_numpy(xq.astype(np.float64)), torch.from_numpy(wq.astype(np.float64)),
                                     padding=1).numpy()
    np.testing.assert_allclose(plain, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    # strided 1x1 and the residual operand
    w1 = (rng.standard_normal((96, 64, 1, 1)) * 0.2).astype(np.float32)
    want = O.pwconv_i8(x, w1, 1, 8, in_max, bn_scale=sc, bn_shift=sh, act="relu", stride=2)
    assert want.shape == (3, 96, 4, 5)
    _eq(H.pwconv_i8(x, w1, 1, 8, in_max=in_max, bn_scale=sc, bn_shift=sh, act="relu", stride=2), want)
    _eq(want, O.pwconv_i8(np.ascontiguousarray(x[:, :, ::2, ::2]), w1, 1, 8, in_max, bn_scale=sc, bn_shift=sh, act="relu"))
    res = (rng.standard_normal((3, 96, 7, 9)) * 3).astype(np.float32)
    for act in ("relu", None):
        want = O.pwconv_i8(x, w1, 1, 8, in_max, bn_scale=sc, bn_shift=sh, act=act, residual=res)
        _eq(H.pwconv_i8(x, w1, 1, 8, in_max=in_max, bn_scale=sc, bn_shift=sh, act=act, residual=res), want)
        two = (O.pwconv_i8(x, w1, 1, 8, in_max, bn_scale=sc, bn_shift=sh) + res).astype(np.float32)
        _eq(want, np.maximum(two, 0) if act == "relu" else two)
    # offline threshold + the batch statistic on the side (host twin: out_current_max still reports the mean)
    y_off = H.pwconv_i8(x, w1, 1, 8, in_max=np.float32(1.5))
    _eq(y_off, O.pwconv_i8(x, w1, 1, 8, np.float32(1.5)))
    # BatchNorm + activation + MaxPool2D(3, 2, 1)
    xb = (rng.standard_normal((2, 5, 9, 12)) * 3).astype(np.float32)
    scb, shb = rng.standard_normal(5).astype(np.float32), rng.standard_normal(5).astype(np.float32)
    for act in ("relu", "none", "relu6"):
        want = O.bn_act_maxpool(xb, scb, shb, act)
        y, st = H.bn_act_maxpool(xb, scb, shb, act, want_stat=True)
        _eq(y, want)
        _eq(st, O.absmax_per_sample(want))
        _eq(want, torch.nn.functional.max_pool2d(torch.from_numpy(O.bn_act(xb, scb, shb, act)), 3, 2, 1).numpy())
    # the 7x7 first convolution
    xs = rng.standard_normal((2, 3, 18, 22)).astype(np.float32)
    w7 = (rng.standard_normal((64, 3, 7, 7)) * 0.1).astype(np.float32)
    sc64, sh64 = rng.uniform(0.5, 1.5, 64).astype(np.float32), rng.standard_normal(64).astype(np.float32)
    got, want = H.stem_conv_s2(xs, w7, None, sc64, sh64, "relu"), O.stem_conv_s2(xs, w7, None, sc64, sh64, "relu")
    assert got.shape == want.shape == (2, 64, 9, 11)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)       # true fmaf vs its fp64 emulation: double-rounding ties
    assert (got != want).mean() < 1e-3


def test_calibration_vs_numpy_oracle():
    rng = np.random.default_rng(5)
    fm = np.maximum(rng.standard_normal((4, 16, 14, 14)), 0).astype(np.float32) * np.float32(3)
    h, m = H.discrete_histogram(fm, 2048)
    wh, wm = O.discrete_histogram(fm, 2048)
    assert m == wm
    _eq(h, wh)
    h300, _ = H.discrete_histogram(fm * np.float32(100), 2048)      # max >= 256: index == bins clamps into the last bin
    _eq(h300, O.discrete_histogram(fm * np.float32(100), 2048)[0])
    for levels in (256, 128, 16):
        assert int(H.kl_search(h, levels, levels)[0]) == O.kl_calibrate(h, levels, levels, 2048)
    with pytest.raises(RuntimeError, match="min_bins should be greater than levels"):
        H.kl_search(h, 256, 128)
    lg = rng.standard_normal((7, 10)).astype(np.float32)
    lg[2, 3] = lg[2, 7] = lg[2].max() + 1                            # tie: first index wins
    lb = np.array([1, 2, 3, 4, 5, 6, 11], np.int64)
    _eq(H.eval_counters(lg, lb), O.eval_counters(lg, lb))
    xc = rng.integers(-128, 128, (64, 64)).astype(np.int8)
    wc = rng.integers(-127, 128, (32, 64)).astype(np.int8)
    _eq(H.gemm_i8_codes(xc, wc, 2, 32, 128), O.gemm_i8_codes(xc, wc, 2, 32, 128))
    _eq(H.ste_forward(lg, np.float32([0.1]), 1.0, -1.0), O.ste_forward(lg, np.float32(0.1), 1.0, -1.0))


def test_thread_count_does_not_change_results():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((6, 16, 28, 28)) * 2).astype(np.float32)
    H.set_threads(1)
    y1, c1, _ = H.fake_quant_online(x)
    h1, _ = H.discrete_histogram(np.abs(x), 2048)
    H.set_threads(0)
    assert H.threads() >= 1
    y2, c2, _ = H.fake_quant_online(x)
    h2, _ = H.discrete_histogram(np.abs(x), 2048)
    assert c1 == c2
    _eq(y1, y2)
    _eq(h1, h2)


def test_sliced_conv3x3_host_twin_equals_numpy_oracle():
    """fq_weight_slices_host / fq_conv3x3_i8_sliced_host (the three-slice form of BASELINE config 5's 3x3 layers) against the
    numpy restatement: digits, power-of-two scales, row sums and the convolution, bit for bit."""
    rng = np.random.default_rng(23)
    for (n, cin, cout, h, w), signed in (((2, 64, 64, 5, 7), False), ((1, 128, 32, 4, 4), True)):
        x = (rng.standard_normal((n, cin, h, w)) * 2).astype(np.float32)
        if not signed:
            x = np.maximum(x, 0)
        wt = O.wino_weight_fake_quant((rng.standard_normal((cout, cin, 3, 3)) * 0.2).astype(np.float32), "F43", 8)[0]
        wt[1] = 0.0
        stat = O.absmax_per_sample(x)
        sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        sh = rng.standard_normal(cout).astype(np.float32)
        (y, st), codes, pscale, rowsum = H.conv3x3_i8_sliced(x, wt, in_stat=stat, signed=signed, bn_scale=sc, bn_shift=sh,
                                                              act="relu", want_stat=True, want_slices=True)
        m, p = O.weight_slices(wt.transpose(0, 2, 3, 1).reshape(cout, -1))
        np.testing.assert_array_equal(pscale, p)
        rows_pad = (cout + 63) // 64 * 64
        for sl, d in enumerate(O.slice_digits(m)):
            np.testing.assert_array_equal(codes[sl, :rows_pad * 9 * cin].reshape(rows_pad, -1)[:cout], d.astype(np.int8))
            np.testing.assert_array_equal(rowsum[sl], d.sum(axis=1).astype(np.int32))
        want = O.conv3x3_i8_sliced(x, wt, O.batch_mean(stat), signed=signed, bn_scale=sc, bn_shift=sh, act="relu")
        np.testing.assert_array_equal(y, want)
        np.testing.assert_array_equal(st, O.absmax_per_sample(want))


def test_c16_hand_over_host_twins_equal_the_fp32_path():
    """fq_pwconv_i8_c16_host / fq_conv3x3_i8_c16_host: a C16 input gives the output of the fp32 input under the same stored
    threshold, a C16 output holds the oracle's codes of the fp32 output (the layout: oracle.to_c16)."""
    rng = np.random.default_rng(31)
    n, cin, cout, h, w = 2, 64, 40, 5, 6
    x = np.maximum(rng.standard_normal((n, cin, h, w)) * 2, 0).astype(np.float32)
    thr, thr2 = np.float32(2.1), np.float32(1.3)
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    sh = rng.standard_normal(cout).astype(np.float32)
    cx = O.ste_codes(x, O.act_scale(thr, False, 8), thr, np.float32(0)).astype(np.int64)
    xc = O.to_c16(cx, 128)
    for kind in ("1x1", "3x3"):
        wt = (rng.standard_normal((cout, cin, 1, 1) if kind == "1x1" else (cout, cin, 3, 3)) * 0.2).astype(np.float32)
        if kind == "1x1":
            want = H.pwconv_i8(x, wt, 1, 8, in_max=thr, bn_scale=sc, bn_shift=sh, act="relu")
            codes, scales = O.weight_codes(wt.reshape(cout, cin), 1, 8)
            cpad = np.zeros((64, 64), np.int8)
            cpad[:cout, :cin] = codes
        else:
            want = H.conv3x3_i8(x, wt, 1, 8, in_max=thr, bn_scale=sc, bn_shift=sh, act="relu")
            codes, scales = O.weight_codes(wt.transpose(0, 2, 3, 1).reshape(cout, -1), 1, 8)
            cpad = np.zeros((64, 9 * cin), np.int8)
            cpad[:cout] = codes
        rowsum = cpad[:cout].astype(np.int32).sum(axis=1).astype(np.int32)
        wantc = O.to_c16(O.ste_codes(want, O.act_scale(thr2, False, 8), thr2, np.float32(0)).astype(np.int64), 128)
        for x_in, is16 in ((x, 0), (xc, 1)):
            for out16 in (False, True):
                if not is16 and not out16:
                    continue
                y = np.zeros(wantc.shape, np.int8) if out16 else np.empty((n, cout, h, w), np.float32)
                common = [scales.astype(np.float32), rowsum, None, y, n, cin]
                tail = [None, np.asarray([thr], np.float32), H._i(8), H._u(0), np.empty(1, np.float32), sc, sh, H._i(1), None]
                othr = np.asarray([thr2], np.float32) if out16 else None
                if kind == "1x1":
                    H._call("fq_pwconv_i8_c16_host", x_in, H._i(is16), cpad, *common, 64, cout, h, w, H._i(1), *tail, None, othr,
                            H._i(8), H._u(0), None, None)
                else:
                    H._call("fq_conv3x3_i8_c16_host", x_in, H._i(is16), cpad, *common, cout, h, w, *tail, othr, H._i(8), H._u(0),
                            None)
                np.testing.assert_array_equal(y, wantc if out16 else want, "%s in16=%d out16=%d" % (kind, is16, out16))


def test_dense_i8_eval_twin_equals_the_two_oracle_steps():
    """fq_dense_i8_eval_host = the oracle's pointwise convolution on planes of one pixel + the oracle's evaluation counters."""
    rng = np.random.default_rng(12)
    n, cin, units = 37, 100, 41
    x = np.maximum(rng.standard_normal((n, cin)) * 2, 0).astype(np.float32)
    w = rng.standard_normal((units, cin)).astype(np.float32)
    w[7] = w[3]                                                       # a tie: the first index wins
    b = rng.standard_normal(units).astype(np.float32)
    b[7] = b[3]
    labels = rng.integers(0, units, n).astype(np.int64)
    labels[0] = units + 1
    stat = O.absmax_per_sample(x)
    y, c = H.dense_i8_eval(x, w, units, 8, labels, None, in_stat=stat, signed=False, width=8, bias=b)
    want = O.pwconv_i8(x.reshape(n, cin, 1, 1), w.reshape(units, cin, 1, 1), units, 8, O.batch_mean(stat), signed=False, width=8,
                       bias=b).reshape(n, units)
    assert np.array_equal(y, want)
    assert np.array_equal(c, O.eval_counters(want, labels, None))


QCONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad, groups
    (2, 2, 5, 5, 10, (3, 3), (1, 1), (1, 1), 1),
    (2, 4, 9, 7, 6, (3, 3), (2, 2), (1, 1), 2),
    (1, 6, 11, 11, 9, (5, 5), (2, 1), (2, 2), 3),
    (3, 8, 8, 8, 8, (3, 3), (2, 2), (1, 1), 8),
    (2, 16, 6, 6, 24, (1, 1), (1, 1), (0, 0), 1),
]


@pytest.mark.parametrize("case", QCONV_CASES, ids=[str(c) for c in QCONV_CASES])
def test_qconv2d_twin_equals_numpy_oracle(golden, case):
    """fq_qconv_weights_prepare_host + fq_qconv2d_forward_host (the twin of the one-call nn.Conv2D(quantized=True) entry
    point) against oracle.qconv2d_forward - itself pinned by golden G8, the reference's own block - over input / weight
    dtypes, bias, integer ReLU, fixed ranges, the producer-statistic range and a folded BatchNorm."""
    n, cin, h, w_, cout, k, st, pad, groups = case
    rng = np.random.default_rng(n * 100 + cin + h)
    x = (rng.standard_normal((n, cin, h, w_)) * 1.5 + 0.4).astype(np.float32)
    w = (rng.standard_normal((cout, cin // groups) + k) * 0.2).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.5).astype(np.float32)
    for in_dt in ("uint8", "int8"):
        for w_dt in ("int8", "uint8"):
            for bias in (None, b):
                for act in (None, "relu"):
                    kw = dict(input_dtype=in_dt, weight_dtype=w_dt, act=act)
                    np.testing.assert_array_equal(H.qconv2d_forward(x, w, bias, st, pad, groups, **kw),
                                                  O.qconv2d_forward(x, w, bias, st, pad, groups, **kw))
    for ir, wr in (((-1.0, 1.0), None), ((0.25, 2.0), (-0.3, 0.3))):
        np.testing.assert_array_equal(H.qconv2d_forward(x, w, b, st, pad, groups, input_range=ir, weight_range=wr),
                                      O.qconv2d_forward(x, w, b, st, pad, groups, input_range=ir, weight_range=wr))
    xr = np.maximum(x, 0)
    stat = xr.reshape(n, -1).max(axis=1)
    bsc, bsh = (rng.random(cout) + 0.5).astype(np.float32), (rng.standard_normal(cout) * 0.1).astype(np.float32)
    for in_dt in (("uint8", "int8") if pad != (0, 0) else ("int8",)):
        kw = dict(input_dtype=in_dt, act="relu", in_stat=stat, bn_scale=bsc, bn_shift=bsh)
        got, gstat = H.qconv2d_forward(xr, w, None, st, pad, groups, want_stat=True, **kw)
        want = O.qconv2d_forward(xr, w, None, st, pad, groups, **kw)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(gstat, np.abs(want).reshape(n, -1).max(axis=1))
        # the statistic of a non-negative tensor IS its range: same result as taking the range from the tensor
        np.testing.assert_array_equal(want, O.qconv2d_forward(xr, w, None, st, pad, groups, input_dtype=in_dt, act="relu",
                                                              bn_scale=bsc, bn_shift=bsh))
    if case == QCONV_CASES[0]:
        g = golden("g8_quantized_conv")
        for use_bias in (0, 1):
            tag = "conv_b%d_g1" % use_bias
            y = H.qconv2d_forward(g[tag + "/x"], g[tag + "/w"], g[tag + "/b"] if use_bias else None, (1, 1), (1, 1), 1)
            np.testing.assert_array_equal(y, g[tag + "/y_int"])


def test_producers_that_bin_what_they_store_equal_the_two_passes():
    """fq_bn_act_stat_hist_host / fq_add_act_stat_hist_host (KL collection, round 4): the plain pass + the histogram of its
    result, against the numpy restatement of the reference's `_discrete_histogram` (distribution_calibrate.py:27-45)."""
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((3, 5, 7, 7)) * 2).astype(np.float32)
    sc, sh = (rng.random(5) + 0.5).astype(np.float32), rng.standard_normal(5).astype(np.float32)
    y0, s0 = H.bn_act(x, sc, sh, "relu", want_stat=True)
    mx_ = np.float32(y0.max() * 0.7)
    y, s, h, neg = H.bn_act_hist(x, sc, sh, "relu", mx_, 2048)
    _eq(y, y0)
    _eq(s, s0)
    _eq(h.astype(np.float32), O.discrete_histogram(y0, 2048, mx_)[0])
    assert neg == 0
    y, s, h, neg = H.bn_act_hist(x, sc, sh, "none", mx_, 64, hist=h[:64].copy())
    assert neg == int((y < 0).sum()) and neg > 0
    b = rng.standard_normal(x.shape).astype(np.float32)
    y1 = np.maximum(x + b, np.float32(0))
    y, s, h, neg = H.add_act_hist(x, b, "relu", np.float32(y1.max()), 128)
    _eq(y, y1)
    _eq(h.astype(np.float32), O.discrete_histogram(y1, 128, np.float32(y1.max()))[0])
    with pytest.raises(RuntimeError):
        H.add_act_hist(x, b, "relu", 1.0, 8192)
