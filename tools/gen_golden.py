#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING THE REFERENCE'S OWN PYTHON on CPU.

Runs only in the build container (needs /root/reference); the GPU box receives the committed .npz files only.
Nothing from /root/reference is copied: the reference modules are imported in place, by path, with bytecode
writing disabled (the tree is read-only by contract).

Two pinning strengths (DESIGN.md "Oracle"):
  * quantize/distribution_calibrate.py imports only numpy + tqdm -> it is loaded unchanged and its results are the
    reference's real results (G1, G2, G3).
  * every other file on the path imports `mxnet`, which cannot be installed here.  Those files are executed
    unchanged on top of `quantization.mxnet_amd.mx` (this project's Gluon-shaped facade, CPU/torch) registered as
    `mxnet`.  That pins the reference's COMPOSITION (op order, epsilon placement, clip bounds, scale formulas,
    broadcasting, the fixed_params state machine); the primitive op semantics are MXNet's documented ones as encoded
    in mx/ndarray.py, not MXNet's binaries (G4..G9).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import sys
sys.dont_write_bytecode = True

import argparse
import importlib
import importlib.util
import os
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED = 7        # examples/simulate_quantization.py:99-100


def install_mxnet_standin():
    from quantization.mxnet_amd import mx
    m = types.ModuleType("mxnet")
    m.nd = mx.nd
    m.ndarray = mx.nd
    m.gluon = mx.gluon
    m.autograd = mx.autograd
    m.initializer = mx.initializer
    m.init = mx.initializer
    m.cpu, m.gpu, m.Context = mx.cpu, mx.gpu, mx.Context
    m.__path__ = []
    sys.modules["mxnet"] = m
    sys.modules["mxnet.nd"] = mx.nd
    sys.modules["mxnet.ndarray"] = mx.nd
    sys.modules["mxnet.gluon"] = mx.gluon
    sys.modules["mxnet.gluon.nn"] = mx.gluon.nn
    sys.modules["mxnet.gluon.data"] = mx.gluon.data
    sys.modules["mxnet.autograd"] = mx.autograd
    sys.modules["mxnet.initializer"] = mx.initializer
    from quantization.mxnet_amd.mx.gluon import parameter as _parameter, block as _block
    from quantization.mxnet_amd.mx import context as _context
    sys.modules["mxnet.gluon.parameter"] = _parameter
    sys.modules["mxnet.gluon.block"] = _block
    sys.modules["mxnet.context"] = _context
    return mx


def load_reference(ref):
    """Import reference sub-packages in place without running quantize/__init__.py (it pulls in the libmxnet-only
    freeze helper)."""
    refq = types.ModuleType("refq")
    refq.__path__ = [os.path.join(ref, "quantize")]
    sys.modules["refq"] = refq
    convert = importlib.import_module("refq.convert")
    initialize = importlib.import_module("refq.initialize")
    spec = importlib.util.spec_from_file_location("refq_distribution_calibrate",
                                                  os.path.join(ref, "quantize", "distribution_calibrate.py"))
    dc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dc)
    refnn = types.ModuleType("refnn")
    refnn.__path__ = [os.path.join(ref, "nn")]
    sys.modules["refnn"] = refnn
    qconv = importlib.import_module("refnn.quantized_conv")
    return convert, initialize, dc, qconv


# ------------------------------------------------------------------------------------------------------
def relu_like(rng, shape, sigma):
    return (np.maximum(rng.standard_normal(shape), 0) * sigma).astype(np.float32)


def gen_hist(dc, out):
    """G1: _discrete_histogram (distribution_calibrate.py:31-47)."""
    rng = np.random.default_rng(SEED)
    cases = {}
    n = 100000
    dists = {
        "halfnormal": np.abs(rng.standard_normal(n)).astype(np.float32) * 1.7,
        "exponential": rng.exponential(0.8, n).astype(np.float32),
        "relu_outlier": np.concatenate([relu_like(rng, (n - 3,), 2.0), np.float32([55.0, 31.5, 80.25])]),
        "tiny_range": (rng.random(5000).astype(np.float32) * np.float32(3e-3)),
        "shape4d": relu_like(rng, (4, 8, 14, 14), 3.0),
    }
    for name, fm in dists.items():
        hist, mx_ = dc._discrete_histogram(fm, 2048, None)
        cases[name + "/fm"] = fm
        cases[name + "/hist_auto"] = hist
        cases[name + "/max_auto"] = np.float32(mx_)
        fixed = np.float32(np.float32(mx_) * np.float32(0.37))
        hist2, mx2 = dc._discrete_histogram(fm, 2048, fixed)
        cases[name + "/max_fixed"] = fixed
        cases[name + "/hist_fixed"] = hist2
        hist3, _ = dc._discrete_histogram(fm, 128, None)
        cases[name + "/hist_auto_b128"] = hist3
    np.savez_compressed(os.path.join(out, "g1_histogram.npz"), **cases)
    return dists


def gen_kl(dc, out, dists):
    """G2: kl_calibrate (distribution_calibrate.py:117-171) + the caller's threshold formula
    (examples/simulate_quantization.py:310)."""
    rng = np.random.default_rng(SEED + 1)
    hists = {}
    for name in ("halfnormal", "exponential", "relu_outlier"):
        h, m = dc._discrete_histogram(dists[name], 2048, None)
        hists[name] = (h, np.float32(m))
    # accumulated multi-batch histogram, a sparse one (many empty bins) and a spiky one
    fm = relu_like(rng, (6, 40000), 1.3)
    h, m = dc._discrete_histogram(fm[0], 2048, None)
    for i in range(1, 6):
        h = h + dc._discrete_histogram(fm[i], 2048, m)[0]
    hists["accumulated6"] = (h, np.float32(m))
    sp = np.zeros(2048, np.float32)
    idx = rng.choice(2048, 300, replace=False)
    sp[idx] = rng.integers(1, 500, 300).astype(np.float32)
    hists["sparse"] = (sp, np.float32(4.0))
    spike = (np.exp(-np.arange(2048) / 90.0) * 4000).astype(np.int32).astype(np.float32)
    spike[1500] += 2500
    hists["spike"] = (spike, np.float32(9.5))
    cases = {}
    for name, (h, m) in hists.items():
        cases[name + "/hist"] = h
        cases[name + "/fm_max"] = m
        for levels in (256, 128, 16, 8):
            best = dc.kl_calibrate(h, levels=levels, min_bins=levels, bins=2048)
            cases["%s/best_L%d" % (name, levels)] = np.int64(best)
            cases["%s/thr_L%d" % (name, levels)] = np.float32((best + 0.5) * (m / 2048))
            print("  kl %-14s levels=%3d -> best_bins=%d" % (name, levels, best), flush=True)
    # small-bins variants (fast to re-check): bins=256
    for name in ("halfnormal", "sparse"):
        h = hists[name][0].reshape(256, 8).sum(axis=1).astype(np.float32)
        cases[name + "/hist_b256"] = h
        for levels in (16, 32):
            cases["%s/best_b256_L%d" % (name, levels)] = np.int64(dc.kl_calibrate(h, levels, levels, 256))
    np.savez_compressed(os.path.join(out, "g2_kl.npz"), **cases)


class _FakeArray(object):
    def __init__(self, a):
        self._a = a

    def asnumpy(self):
        return self._a

    def as_in_context(self, ctx):
        return self


class _FakeHandle(object):
    def __init__(self, blk):
        self._blk = blk

    def detach(self):
        self._blk.hook = None


class _FakeBlock(object):
    def __init__(self):
        self.hook = None

    def register_forward_hook(self, fn):
        self.hook = fn
        return _FakeHandle(self)


class _FakeNet(object):
    """Duck-typed net for collect_feature_maps (distribution_calibrate.py:50-114): block k sees relu(X)*(k+1)."""

    def __init__(self, nblk):
        self.blocks = [_FakeBlock() for _ in range(nblk)]

    def collect_quantized_blocks(self):
        return self.blocks

    def __call__(self, X):
        x = X.asnumpy()
        for k, b in enumerate(self.blocks):
            fm = np.maximum(x, 0) * np.float32(k + 1)
            if k == 2:
                fm = fm[:, :, ::2, ::2]
            if b.hook is not None:
                b.hook(b, (_FakeArray(np.ascontiguousarray(fm)),), None)
        return None


def gen_collect(dc, out):
    """G3: collect_feature_maps over 3 batches (first batch fixes fm_max; later batches clip into it)."""
    rng = np.random.default_rng(SEED + 2)
    batches = [(rng.standard_normal((4, 6, 10, 10)) * s).astype(np.float32) for s in (1.0, 1.6, 0.7)]
    net = _FakeNet(3)
    loader = [(_FakeArray(b), None) for b in batches]
    hists, maxes = dc.collect_feature_maps(net, 2048, loader, ctx=None)
    cases = {"batches": np.stack(batches)}
    for k, b in enumerate(net.blocks):
        cases["hist%d" % k] = hists[b]
        cases["fm_max%d" % k] = np.float32(maxes[b])
    np.savez_compressed(os.path.join(out, "g3_collect.npz"), **cases)


# ------------------------------------------------------------------------------------------------------
def _tie_rich(rng, shape, signed, top):
    """Activations with exact rounding ties, negatives, zeros, values beyond the clip and denormals."""
    x = rng.standard_normal(shape).astype(np.float32) * np.float32(top / 2.5)
    if not signed:
        x = np.where(rng.random(shape) < 0.45, np.float32(0), np.abs(x)).astype(np.float32)
        x.reshape(-1)[::17] *= np.float32(-1)          # a few negatives (clipped to 0 by the unsigned path)
    flat = x.reshape(-1)
    flat[3] = np.float32(1e-41)
    flat[5] = np.float32(-0.0)
    flat[7] = np.float32(top * 3)
    return x


def _capture_conv(mx, convert, initialize, conv, x, **flags):
    """Run the reference's patched forward once; capture what reaches `origin_forward`."""
    cap = {}
    orig = conv.origin_forward

    def spy(F, xq, wq, bias=None):
        cap["xq"], cap["wq"] = xq.asnumpy().copy(), wq.asnumpy().copy()
        cap["bias"] = None if bias is None else bias.asnumpy().copy()
        return orig(F, xq, wq, bias)
    conv.origin_forward = spy
    for k, v in flags.items():
        setattr(conv, k, v)
    y = conv(mx.nd.array(x))
    conv.origin_forward = orig
    cap["y"] = y.asnumpy().copy()
    cap["current_input_max"] = np.float32(conv.current_input_max) if hasattr(conv, "current_input_max") else None
    return cap


def _codes(mx, x, scale, lo, hi):
    """Integer stage of ste_func.py:41 — the reference's expression without the trailing `* scale`."""
    a = mx.nd.array(x)
    return (a.clip(lo, hi) / (scale + 1e-10)).round().asnumpy()


def gen_act(mx, convert, initialize, out):
    """G4: activation branch of _conv2d_forward (convert_conv2d.py:53-66) and _dense_forward (convert_dense.py:39-49)."""
    nn = mx.gluon.nn
    rng = np.random.default_rng(SEED + 3)
    cases = {}
    for shape in ((4, 8, 7, 7), (2, 32, 14, 14), (3, 5, 9, 11)):
        for signed in (False, True):
            for width in (8, 4):
                tag = "conv_%s_%s_w%d" % ("x".join(map(str, shape)), "s" if signed else "u", width)
                x = _tie_rich(rng, shape, signed, 6.0)
                conv = nn.Conv2D(4, 1, in_channels=shape[1], use_bias=False)
                conv.initialize()
                convert.gen_conv2d_converter(input_signed=signed, input_width=width)(conv)
                conv.input_max.initialize(mx.initializer.Constant(0))
                # online
                cap = _capture_conv(mx, convert, initialize, conv, x)
                mx_ = cap["current_input_max"]
                scale = mx_ / (2 ** (width - 1) - 1) if signed else mx_ / (2 ** width - 1)
                lo = -mx_ if signed else 0.0
                cases[tag + "/x"] = x
                cases[tag + "/online_y"] = cap["xq"]
                cases[tag + "/online_max"] = np.float32(mx_)
                cases[tag + "/online_scale"] = np.float32(scale)
                cases[tag + "/online_codes"] = _codes(mx, x, scale, lo, mx_)
                # offline with a stored threshold smaller than the batch statistic
                thr = np.float32(mx_ * np.float32(0.613))
                conv.input_max.set_data(mx.nd.array([thr]))
                cap = _capture_conv(mx, convert, initialize, conv, x, quantize_input_offline=True)
                scale = thr / (2 ** (width - 1) - 1) if signed else thr / (2 ** width - 1)
                lo = -thr if signed else 0.0
                cases[tag + "/offline_thr"] = thr
                cases[tag + "/offline_y"] = cap["xq"]
                cases[tag + "/offline_scale"] = np.float32(scale)
                cases[tag + "/offline_codes"] = _codes(mx, x, scale, lo, thr)
                cases[tag + "/offline_curmax"] = cap["current_input_max"]
    # all-zero input: max_ = 0 -> scale 0 -> divide by fp32(1e-10)
    x0 = np.zeros((2, 3, 4, 4), np.float32)
    conv = nn.Conv2D(2, 1, in_channels=3, use_bias=False)
    conv.initialize()
    convert.gen_conv2d_converter()(conv)
    conv.input_max.initialize(mx.initializer.Constant(0))
    cap = _capture_conv(mx, convert, initialize, conv, x0)
    cases["conv_zero/x"], cases["conv_zero/online_y"] = x0, cap["xq"]
    cases["conv_zero/online_max"] = cap["current_input_max"]
    # dense: clip_min is always 0 even when signed (convert_dense.py:49)
    for signed in (False, True):
        for width in (8, 4):
            tag = "dense_%s_w%d" % ("s" if signed else "u", width)
            x = _tie_rich(rng, (6, 64), signed, 4.0)
            dense = nn.Dense(10, in_units=64)
            dense.initialize()
            convert.gen_dense_converter(input_signed=signed, input_width=width)(dense)
            dense.input_max.initialize(mx.initializer.Constant(0))
            cap = _capture_conv(mx, convert, initialize, dense, x)
            cases[tag + "/x"] = x
            cases[tag + "/online_y"] = cap["xq"]
            cases[tag + "/online_max"] = cap["current_input_max"]
            thr = np.float32(cap["current_input_max"] * np.float32(0.5))
            dense.input_max.set_data(mx.nd.array([thr]))
            cap = _capture_conv(mx, convert, initialize, dense, x, quantize_input_offline=True)
            cases[tag + "/offline_thr"] = thr
            cases[tag + "/offline_y"] = cap["xq"]
    np.savez_compressed(os.path.join(out, "g4_activation.npz"), **cases)


def gen_weight(mx, convert, initialize, out):
    """G5: weight branch (convert_conv2d.py:68-95, convert_dense.py:52-63); G6: Winograd-domain variant (:71-83)."""
    nn = mx.gluon.nn
    rng = np.random.default_rng(SEED + 4)
    cases = {}
    x_for = {}
    specs = {"dw16": dict(channels=16, kernel_size=3, padding=1, groups=16, in_channels=16),
             "pw32x16": dict(channels=32, kernel_size=1, in_channels=16),
             "c8x4k3": dict(channels=8, kernel_size=3, padding=1, in_channels=4)}
    for name, kw in specs.items():
        w_shape = (kw["channels"], kw["in_channels"] // kw.get("groups", 1), kw["kernel_size"], kw["kernel_size"])
        w = (rng.standard_normal(w_shape) * rng.uniform(0.02, 1.5, (w_shape[0], 1, 1, 1))).astype(np.float32)
        w.reshape(-1)[1] = 0.0
        cases[name + "/w"] = w
        x = relu_like(rng, (2, kw["in_channels"], 6, 6), 1.0)
        x_for[name] = x
        for qt in ("layer", "group", "channel"):
            if qt == "group" and kw.get("groups", 1) not in (1, kw["channels"]):
                continue
            for width in (8, 4):
                conv = nn.Conv2D(use_bias=False, **kw)
                conv.initialize()
                conv.weight.set_data(mx.nd.array(w))
                convert.gen_conv2d_converter(weight_width=width, quant_type=qt, quantize_input=False)(conv)
                cap = _capture_conv(mx, convert, initialize, conv, x)
                cases["%s/%s_w%d" % (name, qt, width)] = cap["wq"]
    for qt in ("layer", "channel"):
        for width in (8, 4):
            w = (rng.standard_normal((10, 64)) * rng.uniform(0.05, 1.0, (10, 1))).astype(np.float32)
            dense = nn.Dense(10, in_units=64)
            dense.initialize()
            dense.weight.set_data(mx.nd.array(w))
            convert.gen_dense_converter(weight_width=width, quant_type=qt, quantize_input=False)(dense)
            cap = _capture_conv(mx, convert, initialize, dense, relu_like(rng, (3, 64), 1.0))
            cases["dense/%s_w%d/w" % (qt, width)] = w
            cases["dense/%s_w%d/wq" % (qt, width)] = cap["wq"]
    np.savez_compressed(os.path.join(out, "g5_weight.npz"), **cases)

    cases = {}
    from refq.convert import wino_matrix
    for variant in ("F23", "F43", "F63"):
        G = wino_matrix.Winograd_G[variant].asnumpy()
        cases[variant + "/G"] = G
        cases[variant + "/GI"] = np.linalg.pinv(G)
        cases[variant + "/GTI"] = np.linalg.pinv(G.T)
        for name in ("c8x4k3", "dw16"):
            kw = specs[name]
            w = np.load(os.path.join(out, "g5_weight.npz"))[name + "/w"]
            for width in (8, 4):
                conv = nn.Conv2D(use_bias=False, **kw)
                conv.initialize()
                conv.weight.set_data(mx.nd.array(w))
                convert.gen_conv2d_converter(weight_width=width, quant_type="channel", quantize_input=False,
                                             wino_quantize=variant)(conv)
                cap = _capture_conv(mx, convert, initialize, conv, x_for[name])
                cases["%s/%s_w%d/w" % (variant, name, width)] = w
                cases["%s/%s_w%d/wq" % (variant, name, width)] = cap["wq"]
    np.savez_compressed(os.path.join(out, "g6_winograd.npz"), **cases)


def _tiny_net(mx, rng, signed_dense=False):
    """conv3x3(3->8) [excluded] -> relu -> dw3x3(8) -> relu -> pw1x1(8->12) -> relu -> pool -> dense(12->5)."""
    nn = mx.gluon.nn
    from quantization.mxnet_amd.mx.gluon.block import reset_naming
    reset_naming()
    net = nn.HybridSequential(prefix="tiny_")
    with net.name_scope():
        net.add(nn.Conv2D(8, 3, padding=1, in_channels=3, use_bias=False), nn.Activation("relu"),
                nn.Conv2D(8, 3, padding=1, groups=8, in_channels=8, use_bias=True), nn.Activation("relu"),
                nn.Conv2D(12, 1, in_channels=8, use_bias=False), nn.Activation("relu"),
                nn.GlobalAvgPool2D(), nn.Flatten(), nn.Dense(5, in_units=12))
    net.initialize()
    params = {}
    for name, p in net.collect_params().items():
        a = (rng.standard_normal(p.shape) * 0.4).astype(np.float32)
        p.set_data(mx.nd.array(a))
        params[name] = a
    return net, params


def gen_ema_and_state(mx, convert, initialize, out):
    """G7: _update_ema (convert.py:66-79) over 10 steps from 0.  G9: CLI state machine
    (simulate_quantization.py:250-253,320-323,337-339,346-348; convert_conv2d.py:69,96-105)."""
    rng = np.random.default_rng(SEED + 5)
    cases = {}
    for quant_type, wt in (("layer", 8), ("channel", 4)):
        tag = "%s_w%d" % (quant_type, wt)
        net, params = _tiny_net(mx, rng)
        for k, v in params.items():
            cases["%s/param/%s" % (tag, k)] = v
        convert_fn = {mx.gluon.nn.Conv2D: convert.gen_conv2d_converter(quant_type=quant_type, weight_width=wt),
                      mx.gluon.nn.Dense: convert.gen_dense_converter(quant_type=quant_type, weight_width=wt),
                      mx.gluon.nn.Activation: None, mx.gluon.nn.BatchNorm: None}
        convert.convert_model(net, exclude=[net[0]], convert_fn=convert_fn)
        initialize.qparams_init(net)
        blocks = net.collect_quantized_blocks()
        cases[tag + "/n_blocks"] = np.int64(len(blocks))
        xs = [(rng.standard_normal((4, 3, 8, 8)) * (1 + 0.2 * i)).astype(np.float32) for i in range(10)]
        cases[tag + "/xs"] = np.stack(xs)
        # A. calibration passes: online, update_ema every batch (fixed_params == -1: weights re-quantised each time)
        net.quantize_input(enable=True, online=True)
        ema, cur, logits = [], [], []
        for x in xs:
            y = net(mx.nd.array(x))
            net.update_ema()
            ema.append([b.input_max.data().asscalar() for b in blocks])
            cur.append([np.float32(b.current_input_max) for b in blocks])
            logits.append(y.asnumpy())
        cases[tag + "/calib_ema"] = np.asarray(ema, np.float32)
        cases[tag + "/calib_cur"] = np.asarray(cur, np.float32)
        cases[tag + "/calib_logits"] = np.stack(logits)
        # B. freeze + offline eval
        net.fix_params()
        net.quantize_input(enable=True, online=False)
        cases[tag + "/fixed_before"] = np.asarray([getattr(b, "fixed_params", -9) for b in blocks], np.int64)
        y1 = net(mx.nd.array(xs[0])).asnumpy()
        cases[tag + "/fixed_after"] = np.asarray([getattr(b, "fixed_params", -9) for b in blocks], np.int64)
        y2 = net(mx.nd.array(xs[1])).asnumpy()
        cases[tag + "/offline_logits0"], cases[tag + "/offline_logits1"] = y1, y2
        for name, p in net.collect_params().items():
            cases["%s/frozen/%s" % (tag, name)] = p.data().asnumpy()
        # C. disable_quantize (KL collection mode): frozen weights pass through, inputs untouched
        net.disable_quantize()
        cases[tag + "/disabled_logits"] = net(mx.nd.array(xs[2])).asnumpy()
        net.enable_quantize()
        # D. online again on the frozen net
        net.quantize_input(enable=True, online=True)
        cases[tag + "/online_frozen_logits"] = net(mx.nd.array(xs[3])).asnumpy()
        # E. input quantisation switched off entirely
        net.quantize_input(enable=False)
        cases[tag + "/noinput_logits"] = net(mx.nd.array(xs[4])).asnumpy()
    np.savez_compressed(os.path.join(out, "g7_g9_ema_state.npz"), **cases)


def gen_qconv(mx, qconv, convert, initialize, out):
    """G8: nn/quantized_conv.py — _quantize/quantize/dequantize (:54-76) and the (2,2,5,5)->10ch Conv2D case of the
    reference's tests/test_quantized_conv.py:36-41, now seeded; three-way: int-code conv / simulated conv / float."""
    F = mx.nd
    nn = mx.gluon.nn
    rng = np.random.default_rng(SEED + 6)
    cases = {}
    for name, a in (("u01", rng.random((2, 2, 5, 5)).astype(np.float32)),
                    ("normal", rng.standard_normal((3, 4, 6, 6)).astype(np.float32)),
                    ("shifted", (rng.random((2, 3, 4, 4)) * 3 + 0.5).astype(np.float32))):
        cases[name + "/x"] = a
        for t in ("int8", "uint8"):
            codes, scale = qconv.quantize(F, mx.nd.array(a), t)
            cases["%s/%s_codes" % (name, t)] = codes.asnumpy()
            cases["%s/%s_scale" % (name, t)] = np.float32(scale)
            cases["%s/%s_deq" % (name, t)] = qconv.dequantize(F, codes, scale).asnumpy()
    for use_bias in (False, True):
        for groups in (1, 2):
            tag = "conv_b%d_g%d" % (int(use_bias), groups)
            x = rng.random((2, 2, 5, 5)).astype(np.float32)
            w = (rng.standard_normal((10, 2 // groups, 3, 3)) * 0.5).astype(np.float32)
            b = (rng.standard_normal(10) * 0.3).astype(np.float32)
            cases[tag + "/x"], cases[tag + "/w"], cases[tag + "/b"] = x, w, b
            for quantized in (False, True):
                c = qconv.Conv2D(10, 3, 1, 1, in_channels=2, groups=groups, use_bias=use_bias, quantized=quantized,
                                 input_dtype="uint8", weight_dtype="int8")
                c.initialize()
                c.weight.set_data(mx.nd.array(w))
                if use_bias:
                    c.bias.set_data(mx.nd.array(b))
                cases[tag + ("/y_int" if quantized else "/y_float")] = c(mx.nd.array(x)).asnumpy()
            sim = nn.Conv2D(10, 3, 1, 1, groups=groups, in_channels=2, use_bias=use_bias)
            sim.initialize()
            sim.weight.set_data(mx.nd.array(w))
            if use_bias:
                sim.bias.set_data(mx.nd.array(b))
            convert.gen_conv2d_converter()(sim)
            sim.input_max.initialize(mx.initializer.Constant(0))
            cases[tag + "/y_sim"] = sim(mx.nd.array(x)).asnumpy()
    np.savez_compressed(os.path.join(out, "g8_quantized_conv.npz"), **cases)


def _bn_net(mx, rng):
    """conv3x3(3->8) bn relu | dw3x3(8) bn relu | pw1x1(8->12) bn relu | pool | dense(12->5), gluon-style names."""
    nn = mx.gluon.nn
    from quantization.mxnet_amd.mx.gluon.block import reset_naming
    reset_naming()
    net = nn.HybridSequential(prefix="bnnet_")
    with net.name_scope():
        net.add(nn.Conv2D(8, 3, padding=1, in_channels=3, use_bias=False), nn.BatchNorm(in_channels=8), nn.Activation("relu"),
                nn.Conv2D(8, 3, padding=1, groups=8, in_channels=8, use_bias=False), nn.BatchNorm(in_channels=8),
                nn.Activation("relu"),
                nn.Conv2D(12, 1, in_channels=8, use_bias=True), nn.BatchNorm(in_channels=12), nn.Activation("relu"),
                nn.GlobalAvgPool2D(), nn.Flatten(), nn.Dense(5, in_units=12))
    net.initialize()
    params = {}
    for name, p in net.collect_params().items():
        if name.endswith("running_var"):
            a = rng.uniform(0.5, 2.0, p.shape).astype(np.float32)
        elif name.endswith("gamma"):
            a = rng.uniform(0.5, 1.5, p.shape).astype(np.float32)
        else:
            a = (rng.standard_normal(p.shape) * 0.4).astype(np.float32)
        p.set_data(mx.nd.array(a))
        params[name] = a
    return net, params


def gen_fake_bn(mx, convert, initialize, ref, out):
    """G10: fake-BN fold (convert_conv2d.py:47-51,122-141; initialize.py:46-70; convert_bn.py) = the CLI's --merge-bn,
    and the one-shot `merge_bn` (quantize/freeze/merge_bn.py:34-92, loaded by path)."""
    spec = importlib.util.spec_from_file_location("refq_merge_bn", os.path.join(ref, "quantize", "freeze", "merge_bn.py"))
    mb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mb)
    nn = mx.gluon.nn
    rng = np.random.default_rng(SEED + 7)
    cases = {}
    xs = [(rng.standard_normal((4, 3, 8, 8)) * 1.3).astype(np.float32) for _ in range(3)]
    cases["xs"] = np.stack(xs)
    # --- A: CLI --merge-bn flow -------------------------------------------------------------------------------
    net, params = _bn_net(mx, rng)
    for k, v in params.items():
        cases["param/" + k] = v
    convert_fn = {nn.Conv2D: convert.gen_conv2d_converter(fake_bn=True, input_signed=True),
                  nn.Dense: convert.gen_dense_converter(input_signed=True),
                  nn.Activation: None, nn.BatchNorm: convert.bypass_bn}
    convert.convert_model(net, exclude=[net[0], net[1]], convert_fn=convert_fn)
    initialize.qparams_init(net)
    net.quantize_input(enable=True, online=True)
    cases["fakebn/calib_logits"] = net(mx.nd.array(xs[0])).asnumpy()
    net.fix_params()
    cases["fakebn/frozen_logits0"] = net(mx.nd.array(xs[1])).asnumpy()
    cases["fakebn/frozen_logits1"] = net(mx.nd.array(xs[2])).asnumpy()
    for name, p in net.collect_params().items():
        cases["fakebn/frozen/" + name] = p.data().asnumpy()
    # --- B: merge_bn on a fresh float net ---------------------------------------------------------------------------
    rng2 = np.random.default_rng(SEED + 7)
    _ = [(rng2.standard_normal((4, 3, 8, 8)) * 1.3) for _ in range(3)]
    net2, _p = _bn_net(mx, rng2)
    before = net2(mx.nd.array(xs[0])).asnumpy()
    mb.merge_bn(net2)
    cases["merge/logits_before"] = before
    cases["merge/logits_after"] = net2(mx.nd.array(xs[0])).asnumpy()
    for name, p in net2.collect_params().items():
        cases["merge/param/" + name] = p.data().asnumpy()
    np.savez_compressed(os.path.join(out, "g10_fake_bn.npz"), **cases)


def _qat_net(mx, params):
    """conv3x3(3->8) bn relu | dw3x3 s2 (8) bn relu | pw1x1(8->16) bn relu | pool | dense(16->10) - the net of
    tests/test_qat.py, parameters given."""
    nn = mx.gluon.nn
    from quantization.mxnet_amd.mx.gluon.block import reset_naming
    reset_naming()
    net = nn.HybridSequential()
    net.add(nn.Conv2D(8, 3, 1, 1, use_bias=False, in_channels=3), nn.BatchNorm(in_channels=8), nn.Activation("relu"),
            nn.Conv2D(8, 3, 2, 1, groups=8, use_bias=False, in_channels=8), nn.BatchNorm(in_channels=8),
            nn.Activation("relu"),
            nn.Conv2D(16, 1, 1, 0, use_bias=False, in_channels=8), nn.BatchNorm(in_channels=16), nn.Activation("relu"),
            nn.GlobalAvgPool2D(), nn.Flatten(), nn.Dense(10, in_units=16))
    net.initialize()
    kids = list(net._children.values())
    A = lambda a: mx.nd.array(np.array(a, copy=True))          # (nd.array may alias the numpy buffer; BatchNorm updates in place)
    for i, ci in enumerate((0, 3, 6)):
        kids[ci].weight.set_data(A(params["c%d_w" % i]))
        bn = kids[ci + 1]
        bn.gamma.set_data(A(params["b%d_g" % i]))
        bn.beta.set_data(A(params["b%d_b" % i]))
        bn.running_mean.set_data(A(params["b%d_m" % i]))
        bn.running_var.set_data(A(params["b%d_v" % i]))
    kids[11].weight.set_data(A(params["d_w"]))
    kids[11].bias.set_data(A(params["d_b"]))
    return net, kids


def gen_qat(mx, convert, initialize, out):
    """G11: the quantisation-aware-training step made by the REFERENCE'S OWN code under `autograd.record()`:
    `convert_model` with its converters (`LinearQuantizeSTE.forward/backward`, ste_func.py:30-44; the fake-BN fold and its
    batch-statistic pre-hook `_add_fake_bn_ema_hook`, convert_conv2d.py:47-51,144-154), `net.update_ema()`
    (convert.py:66-79) - the notebook's loop body (examples/quantize_aware_training_cifar10.ipynb cell 15) WITHOUT the
    optimiser: the parameters stay put, every step sees a new batch, so nothing in the fixture depends on this project's
    Trainer.  What the stand-in supplies: convolution / dense / BatchNorm(train) / softmax-cross-entropy and their
    gradients (torch CPU) and the tape.  Per step: per-sample loss, logits, the gradient of every parameter, and after
    `update_ema` every `input_max` and every moving statistic.
      A  ordinary BatchNorm net, W8A8 per layer and per channel, online for two steps, OFFLINE in the third
      B  the notebook's configuration (cells 6-7): per-channel W4A4, fake_bn=True, BatchNorm bypassed, first conv + BN
         excluded, input quantisers off for two steps, then on OFFLINE with the EMA'd thresholds"""
    nn = mx.gluon.nn
    autograd, gluon = mx.autograd, mx.gluon
    rng = np.random.default_rng(SEED + 8)
    params = {"c0_w": rng.standard_normal((8, 3, 3, 3)) * 0.4, "c1_w": rng.standard_normal((8, 1, 3, 3)) * 0.4,
              "c2_w": rng.standard_normal((16, 8, 1, 1)) * 0.4, "d_w": rng.standard_normal((10, 16)) * 0.4,
              "d_b": rng.standard_normal(10) * 0.1}
    for i, c in enumerate((8, 8, 16)):
        params["b%d_g" % i] = rng.uniform(0.5, 1.5, c)
        params["b%d_b" % i] = rng.standard_normal(c) * 0.1
        params["b%d_m" % i] = rng.standard_normal(c) * 0.1
        params["b%d_v" % i] = rng.uniform(0.5, 1.5, c)
    params = {k: np.asarray(v, dtype=np.float32) for k, v in params.items()}
    steps = 4
    Xs = [rng.standard_normal((6, 3, 12, 12)).astype(np.float32) * np.float32(1.0 + 0.3 * s) for s in range(steps)]
    ys = [rng.integers(0, 10, 6).astype(np.float32) for _ in range(steps)]
    cases = {"steps": np.int64(steps), "xs": np.stack(Xs), "ys": np.stack(ys)}
    for k, v in params.items():
        cases["param/" + k] = v.copy()
    loss_func = gluon.loss.SoftmaxCrossEntropyLoss()

    def run(tag, net, named, offline_at, enable_before):
        blocks = net.collect_quantized_blocks()
        for s in range(steps):
            if s == offline_at:
                net.quantize_input(enable=True, online=False)
            for p in net.collect_params().values():              # fresh gradients every step (grad_req 'write')
                if p._data is not None and p.data()._t.grad is not None:
                    p.data()._t.grad = None
            with autograd.record():
                outputs = net(mx.nd.array(Xs[s]))
                loss = loss_func(outputs, mx.nd.array(ys[s]))
            net.update_ema()
            loss.backward()
            cases["%s/step%d/loss" % (tag, s)] = loss.asnumpy()
            cases["%s/step%d/logits" % (tag, s)] = outputs.asnumpy()
            for k, p in named.items():
                g = p.data()._t.grad
                if g is not None:
                    cases["%s/step%d/grad/%s" % (tag, s, k)] = g.detach().numpy().copy()
                cases["%s/step%d/value/%s" % (tag, s, k)] = p.data().asnumpy().copy()
            cases["%s/step%d/input_max" % (tag, s)] = np.asarray(
                [b.input_max.data().asnumpy()[0] for b in blocks if getattr(b, "input_max", None) is not None], np.float32)

    for qt in ("layer", "channel"):
        net, kids = _qat_net(mx, params)
        convert.convert_model(net, convert_fn={nn.Conv2D: convert.gen_conv2d_converter(quant_type=qt),
                                               nn.Dense: convert.gen_dense_converter(quant_type=qt),
                                               nn.Activation: None, nn.BatchNorm: None})
        initialize.qparams_init(net)
        net.quantize_input(enable=True, online=True)
        named = {"c0_w": kids[0].weight, "c1_w": kids[3].weight, "c2_w": kids[6].weight, "d_w": kids[11].weight,
                 "d_b": kids[11].bias}
        for i, ci in enumerate((1, 4, 7)):
            named.update({"b%d_g" % i: kids[ci].gamma, "b%d_b" % i: kids[ci].beta, "b%d_m" % i: kids[ci].running_mean,
                          "b%d_v" % i: kids[ci].running_var})
        run("bn_" + qt, net, named, offline_at=2, enable_before=True)

    nb = dict(params)
    nb["c1_b"] = (rng.standard_normal(8) * 0.05).astype(np.float32)
    nb["c2_b"] = (rng.standard_normal(16) * 0.05).astype(np.float32)
    cases["param/c1_b"], cases["param/c2_b"] = nb["c1_b"].copy(), nb["c2_b"].copy()
    reset = __import__("quantization.mxnet_amd.mx.gluon.block", fromlist=["reset_naming"]).reset_naming
    reset()
    net = nn.HybridSequential()
    net.add(nn.Conv2D(8, 3, 1, 1, use_bias=False, in_channels=3), nn.BatchNorm(in_channels=8), nn.Activation("relu"),
            nn.Conv2D(8, 3, 2, 1, groups=8, use_bias=True, in_channels=8), nn.BatchNorm(in_channels=8),
            nn.Activation("relu"),
            nn.Conv2D(16, 1, 1, 0, use_bias=True, in_channels=8), nn.BatchNorm(in_channels=16), nn.Activation("relu"),
            nn.GlobalAvgPool2D(), nn.Flatten(), nn.Dense(10, in_units=16))
    net.initialize()
    kids = list(net._children.values())
    A = lambda a: mx.nd.array(np.array(a, copy=True))
    for i, ci in enumerate((0, 3, 6)):
        kids[ci].weight.set_data(A(nb["c%d_w" % i]))
        if i:
            kids[ci].bias.set_data(A(nb["c%d_b" % i]))
        bn = kids[ci + 1]
        bn.gamma.set_data(A(nb["b%d_g" % i]))
        bn.beta.set_data(A(nb["b%d_b" % i]))
        bn.running_mean.set_data(A(nb["b%d_m" % i]))
        bn.running_var.set_data(A(nb["b%d_v" % i]))
    kids[11].weight.set_data(A(nb["d_w"]))
    kids[11].bias.set_data(A(nb["d_b"]))
    converter = {nn.Conv2D: convert.gen_conv2d_converter(quant_type="channel", fake_bn=True, input_width=4, weight_width=4),
                 nn.Dense: convert.gen_dense_converter(quant_type="channel", input_width=4, weight_width=4),
                 nn.Activation: None, nn.BatchNorm: convert.bypass_bn}
    convert.convert_model(net, exclude=[kids[0], kids[1]], convert_fn=converter)
    net.quantize_input(enable=False)
    initialize.qparams_init(net)
    named = {"c0_w": kids[0].weight, "b0_g": kids[1].gamma, "b0_b": kids[1].beta, "b0_m": kids[1].running_mean,
             "b0_v": kids[1].running_var, "d_w": kids[11].weight, "d_b": kids[11].bias}
    for i, ci in ((1, 3), (2, 6)):
        c = kids[ci]
        named.update({"c%d_w" % i: c.weight, "c%d_b" % i: c.bias, "b%d_g" % i: c.gamma, "b%d_b" % i: c.beta,
                      "b%d_m" % i: c.running_mean, "b%d_v" % i: c.running_var})
    run("notebook", net, named, offline_at=2, enable_before=False)
    np.savez_compressed(os.path.join(out, "g11_qat.npz"), **cases)


def gen_dense_unflattened(mx, convert, initialize, out):
    """G12 (round 6): `_dense_forward` (convert_dense.py:39-49) on an UN-flattened (N, C, H, W) input - vgg's first Dense behind a
    pooling layer: `F.max(F.abs(x), axis=1).mean()` reduces over C only.  The reference's own forward through the stand-in."""
    nn = mx.gluon.nn
    rng = np.random.default_rng(SEED + 12)
    cases = {}
    for shape in ((3, 8, 2, 2), (2, 16, 7, 7), (4, 6, 1, 3)):
        for signed in (False, True):
            tag = "dense4d_%s_%s" % ("x".join(map(str, shape)), "s" if signed else "u")
            x = _tie_rich(rng, shape, signed, 4.0)
            dense = nn.Dense(5, in_units=int(np.prod(shape[1:])))
            dense.initialize()
            convert.gen_dense_converter(input_signed=signed, input_width=8)(dense)
            dense.input_max.initialize(mx.initializer.Constant(0))
            cap = _capture_conv(mx, convert, initialize, dense, x)
            cases[tag + "/x"] = x
            cases[tag + "/online_y"] = cap["xq"]
            cases[tag + "/online_max"] = cap["current_input_max"]
            thr = np.float32(cap["current_input_max"] * np.float32(0.5))
            dense.input_max.set_data(mx.nd.array([thr]))
            cap = _capture_conv(mx, convert, initialize, dense, x, quantize_input_offline=True)
            cases[tag + "/offline_thr"] = thr
            cases[tag + "/offline_y"] = cap["xq"]
            cases[tag + "/offline_curmax"] = cap["current_input_max"]
    np.savez_compressed(os.path.join(out, "g12_dense_unflattened.npz"), **cases)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    np.random.seed(SEED)
    mx = install_mxnet_standin()
    convert, initialize, dc, qconv = load_reference(args.ref)
    only = set(filter(None, args.only.split(",")))

    def want(k):
        return not only or k in only
    if want("g1") or want("g2"):
        dists = gen_hist(dc, args.out)
        print("g1 done", flush=True)
    if want("g2"):
        gen_kl(dc, args.out, dists)
        print("g2 done", flush=True)
    if want("g3"):
        gen_collect(dc, args.out)
        print("g3 done", flush=True)
    if want("g4"):
        gen_act(mx, convert, initialize, args.out)
        print("g4 done", flush=True)
    if want("g5"):
        gen_weight(mx, convert, initialize, args.out)
        print("g5/g6 done", flush=True)
    if want("g7"):
        gen_ema_and_state(mx, convert, initialize, args.out)
        print("g7/g9 done", flush=True)
    if want("g8"):
        gen_qconv(mx, qconv, convert, initialize, args.out)
        print("g8 done", flush=True)
    if want("g10"):
        gen_fake_bn(mx, convert, initialize, args.ref, args.out)
        print("g10 done", flush=True)
    if want("g11"):
        gen_qat(mx, convert, initialize, args.out)
        print("g11 done", flush=True)
    if want("g12"):
        gen_dense_unflattened(mx, convert, initialize, args.out)
        print("g12 done", flush=True)
    with open(os.path.join(args.out, "PROVENANCE.txt"), "w") as f:
        import scipy
        import torch
        f.write("generated by tools/gen_golden.py from the reference at %s\n" % args.ref)
        f.write("python %s\nnumpy %s\ntorch %s\nscipy %s\nseed %d\n"
                % (sys.version.split()[0], np.__version__, torch.__version__, scipy.__version__, SEED))
        f.write("numpy-promotion hazard: fixtures reflect numpy 2.x (NEP 50) scalar promotion; see SURVEY.md 8c\n")
    for d in (os.path.join(args.ref, "quantize"), os.path.join(args.ref, "nn")):
        for dirpath, dirnames, _ in os.walk(d):
            assert "__pycache__" not in dirnames, "bytecode leaked into the reference tree: " + dirpath


if __name__ == "__main__":
    main()
