"""GPU parity of the recompute pair (round 6): fq_pwconv_i8_stat + fq_pwdw_fused against
  (a) the two launches they replace - fq_pwconv_i8 then fq_dwconv3x3 - bit for bit (outputs, per-sample statistics, both
      `current_input_max` scalars), and
  (b) the host twins of those two (oracle/fq_host.cpp), the full-size oracle of the net tests.
Reference: the chain `Conv2D 1x1 -> BatchNorm -> ReLU -> [converted depthwise Conv2D: activation branch
quantize/convert/convert_conv2d.py:53-66 + F.Convolution :108] -> BatchNorm -> ReLU` of the model zoo's MobileNets."""
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


# (n, cin, cout, h, w, stride)
CASES = [
    (2, 32, 64, 112, 112, 2),      # MobileNet1.0 pw1 -> dw2: one slab, four strips, two channel tiles per wavefront
    (2, 64, 128, 56, 56, 1),       # pw2 -> dw3: stride 1, two strips x two channel groups
    (2, 128, 128, 56, 56, 2),      # pw3 -> dw4
    (3, 128, 256, 28, 28, 1),      # pw4 -> dw5: one strip narrower than 30 columns (left zero column at p = 2)
    (2, 256, 256, 28, 28, 2),      # pw5 -> dw6: eight slabs
    (2, 16, 32, 112, 112, 2),      # MobileNet0.5's first pair: half a slab, one channel tile per wavefront
    (2, 32, 64, 40, 36, 1),        # last strip anchored over its neighbour, a partial last band (40 = 14 + 14 + 12 rows)
    (2, 32, 32, 30, 32, 2),        # two strips of a 32-wide plane, stride 2, 15 output rows in two bands
    (1, 64, 64, 18, 60, 1),        # exactly two strips
    (5, 32, 64, 8, 8, 1),          # a plane of one band and one strip (left zero column at p = 22 is not instantiated: refused)
]
MODES = ["online_u8_bn_relu", "online_u8_bn_relu6", "online_u8_bias_nobn", "online_s8_lo_neg", "offline_u8_bn_relu",
         "online_u8_w4_channel"]


def _make(case, mode, dev, ops, seed=0):
    n, cin, cout, h, w, stride = case
    rng = np.random.default_rng(seed + 17 * cin + h)
    signed = "s8" in mode
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32) * np.float32(1.7)
    if not signed:
        x = np.maximum(x, 0)
    x[0, 0, 0, 0] = np.float32(5.5)
    w1 = (rng.standard_normal((cout, cin, 1, 1)) * 0.2).astype(np.float32)
    w2 = (rng.standard_normal((cout, 1, 3, 3)) * 0.3).astype(np.float32)
    k = dict(case=case, signed=signed, wt_width=4 if "w4" in mode else 8, rps=1 if "channel" in mode else cout)
    k["act1"] = k["act2"] = "relu6" if "relu6" in mode else ("relu" if "relu" in mode or "lo_neg" in mode else None)
    if "nobn" in mode:
        k["bn1"] = k["bn2"] = None
        k["b1"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        k["b2"] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        k["act1"], k["act2"] = "relu", None
    else:
        k["bn1"] = ((0.5 + rng.random(cout)).astype(np.float32) * np.where(rng.random(cout) < 0.1, -1, 1).astype(np.float32),
                    (rng.standard_normal(cout) * 0.3).astype(np.float32))
        k["bn2"] = ((0.5 + rng.random(cout)).astype(np.float32), (rng.standard_normal(cout) * 0.3).astype(np.float32))
        k["b1"] = k["b2"] = None
    if "lo_neg" in mode:
        k["act1"] = None              # signed values reach the depthwise layer: its clip range is [-max, max]
    k["offline"] = "offline" in mode
    k["x"], k["w1"], k["w2"] = x, w1, w2
    return k


def _t(a, dev):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _run_pair(k, dev, ops, fused):
    n, cin, cout, h, w, stride = k["case"]
    x = _t(k["x"], dev)
    flags = ops.act_flags(signed=k["signed"])
    codes, scales, rowsum = ops.weight_codes(_t(k["w1"], dev).reshape(cout, cin), k["rps"], k["wt_width"])
    w2 = ops.weight_fake_quant(_t(k["w2"], dev), cout, 8)
    bn1 = (None, None) if k["bn1"] is None else (_t(k["bn1"][0], dev), _t(k["bn1"][1], dev))
    bn2 = (None, None) if k["bn2"] is None else (_t(k["bn2"][0], dev), _t(k["bn2"][1], dev))
    b1, b2 = _t(k["b1"], dev), _t(k["b2"], dev)
    cur1 = torch.zeros(1, device=dev)
    cur2 = torch.zeros(1, device=dev)
    xstat = ops.absmax_per_sample(x)
    if k["offline"]:
        thr1 = torch.full((1,), 4.25, device=dev)
        thr2 = torch.full((1,), 3.5, device=dev)
        in1 = dict(in_thr=thr1, in_stat=xstat)
    else:
        thr2 = None
        in1 = dict(in_stat=xstat)
    if not fused:
        y, ystat = ops.pwconv_i8(x, codes, scales, rowsum, b1, width=8, flags=flags, cur_out=cur1, bn_scale=bn1[0],
                                 bn_shift=bn1[1], act=k["act1"], **in1)
        in2 = dict(in_thr=thr2, in_stat=ystat) if k["offline"] else dict(in_stat=ystat)
        z, zstat = ops.dwconv3x3(y, w2, b2, stride=stride, width=8, flags=flags, cur_out=cur2, bn_scale=bn2[0],
                                 bn_shift=bn2[1], act=k["act2"], **in2)
        return dict(z=N(z), zstat=N(zstat), ystat=N(ystat), cur1=N(cur1), cur2=N(cur2))
    ystat = ops.pwconv_i8_stat(x, codes, scales, rowsum, b1, width=8, flags=flags, cur_out=cur1, bn_scale=bn1[0],
                               bn_shift=bn1[1], act=k["act1"], **in1)
    z, zstat = ops.pwdw_fused(x, codes, scales, rowsum, w2, pw_bias=b1, width=8, flags=flags, pw_bn_scale=bn1[0],
                              pw_bn_shift=bn1[1], pw_act=k["act1"], mid_stat=ystat, mid_thr=thr2, mid_width=8,
                              mid_flags=flags, mid_cur_out=cur2, dw_bias=b2, stride=stride, dw_bn_scale=bn2[0],
                              dw_bn_shift=bn2[1], dw_act=k["act2"], in_stat=None if k["offline"] else xstat,
                              in_thr=in1.get("in_thr"))
    return dict(z=N(z), zstat=N(zstat), ystat=N(ystat), cur1=N(cur1), cur2=N(cur2))


def _host_pair(k, ops, dev):
    from oracle import host as H
    n, cin, cout, h, w, stride = k["case"]
    x = k["x"]
    xstat = H.absmax_per_sample(x)
    bn1 = (None, None) if k["bn1"] is None else k["bn1"]
    bn2 = (None, None) if k["bn2"] is None else k["bn2"]
    off = k["offline"]
    y, ystat = H.pwconv_i8(x, k["w1"], k["rps"], k["wt_width"], in_max=4.25 if off else None, in_stat=xstat, signed=k["signed"],
                           bias=k["b1"], bn_scale=bn1[0], bn_shift=bn1[1], act=k["act1"], want_stat=True)
    w2 = N(ops.weight_fake_quant(_t(k["w2"], dev), cout, 8))
    z, zstat = H.dwconv3x3(y, w2, k["b2"], stride, in_max=3.5 if off else None, in_stat=ystat, signed=k["signed"],
                           bn_scale=bn2[0], bn_shift=bn2[1], act=k["act2"], want_stat=True)
    return dict(z=z, zstat=zstat, ystat=ystat)


@pytest.mark.parametrize("case", CASES, ids=["%dx%d->%d@%dx%ds%d" % c for c in CASES])
@pytest.mark.parametrize("mode", MODES)
def test_pwdw_fused_equals_the_two_launches_and_the_host_twins(dev, ops, case, mode):
    n, cin, cout, h, w, stride = case
    if not ops.pwdw_supported((n, cin, h, w), cout, stride):
        assert (h, w) == (8, 8), "an instantiated shape is refused: %s" % (case,)
        with pytest.raises(Exception):
            _run_pair(_make(case, mode, dev, ops), dev, ops, fused=True)
        return
    k = _make(case, mode, dev, ops)
    two = _run_pair(k, dev, ops, fused=False)
    one = _run_pair(k, dev, ops, fused=True)
    _eq(one["ystat"], two["ystat"], "statistic-only pass vs the storing pointwise kernel")
    _eq(one["cur1"], two["cur1"], "current_input_max of the 1x1 block")
    _eq(one["cur2"], two["cur2"], "current_input_max of the depthwise block")
    _eq(one["z"], two["z"], "fused output vs the two launches")
    _eq(one["zstat"], two["zstat"], "fused per-sample statistic vs the two launches")
    assert np.abs(one["z"]).max() > 0
    host = _host_pair(k, ops, dev)
    _eq(one["ystat"], host["ystat"], "statistic-only pass vs host twin")
    _eq(one["z"], host["z"], "fused output vs host twins")
    _eq(one["zstat"], host["zstat"], "fused statistic vs host twins")


def test_pwdw_fused_full_batch_properties(dev, ops):
    """MobileNet1.0's first pair at the BASELINE batch (128, 32 -> 64 @112x112, stride 2): bit-equal to the two launches."""
    case = (128, 32, 64, 112, 112, 2)
    k = _make(case, "online_u8_bn_relu", dev, ops, seed=5)
    two = _run_pair(k, dev, ops, fused=False)
    one = _run_pair(k, dev, ops, fused=True)
    for name in ("ystat", "cur1", "cur2", "z", "zstat"):
        _eq(one[name], two[name], name)


def test_fast_quotient_is_the_ieee_quotient_where_a_code_depends_on_it(dev, ops):
    """The recompute kernels divide with q = c * y, r = fma(-q, d, c), q' = fma(r, y, q), y = RN(1 / d) (csrc/fq_common.h:
    fast_quot) instead of the fp64-reciprocal form.  Stress it where a code could flip: dividends 0, +-1 ... +-4 ulp around
    every k + 0.5 rounding tie of the quotient, for thousands of divisors scale + 1e-10 (thresholds from 1e-12 to 3e2, widths 2
    to 16), divisors with special significands (powers of two, all ones - which must take the other path - and their
    neighbours), random dividends, zero and denormal dividends.  Bar: the fp32 quotient itself is the IEEE quotient bit for bit
    for every dividend whose quotient is at least 0.25 (below that every code is 0 whatever the last bits)."""
    from quantization.mxnet_amd import _lib
    import ctypes
    rng = np.random.default_rng(2026)
    divisors = []
    for trial in range(1500):
        width = int(rng.choice([8, 8, 8, 4, 2, 7, 16]))
        levels = 2 ** width - 1
        thr = np.float32(10.0 ** rng.uniform(-12, 2.5))
        divisors.append((np.float32(np.float32(thr / np.float32(levels)) + np.float32(1e-10)), levels))
    for e in (-20, -3, 0, 5):                                  # powers of two, all-ones significands and their neighbours
        one = np.float32(2.0 ** e)
        allones = np.nextafter(np.float32(2.0 ** (e + 1)), np.float32(0))
        for d in (one, np.nextafter(one, np.float32(np.inf)), allones, np.nextafter(allones, np.float32(0))):
            divisors.append((np.float32(d), 255))
    took = torch.zeros(1, dtype=torch.int32, device=dev)
    fast_paths = 0
    for d, levels in divisors:
        ks = np.arange(0, min(levels, 4096) + 2, dtype=np.float64)
        base = ((ks + 0.5) * np.float64(d)).astype(np.float32)
        xs = [base]
        for _ in range(4):
            xs = [np.nextafter(xs[0], np.float32(np.inf))] + xs + [np.nextafter(xs[-1], np.float32(-np.inf))]
        c = np.concatenate(xs + [(rng.random(512) * float(d) * levels).astype(np.float32),
                                 np.float32([0.0, 1e-45, 1e-39, float(d), float(d) * levels])]).astype(np.float32)
        ct = torch.from_numpy(c).to(dev)
        out = torch.empty_like(ct)
        dt = torch.from_numpy(np.float32([d])).to(dev)
        _lib.check_call(_lib.LIB.fq_debug_fast_quotient(ctypes.c_void_p(ct.data_ptr()), c.size, ctypes.c_void_p(dt.data_ptr()),
                                                        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(took.data_ptr()),
                                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        want = (c / d).astype(np.float32)                       # numpy's fp32 division is the IEEE one
        got = N(out)
        sel = want >= np.float32(0.25)
        _eq(got[sel], want[sel], "quotients by %r" % (d,))
        assert np.array_equal(np.round(got[~sel]), np.zeros((~sel).sum())), "small quotients by %r round to 0" % (d,)
        fast = int(N(took)[0])
        fast_paths += fast
        if (np.float32(d).view(np.uint32) & 0x7FFFFF) == 0x7FFFFF:
            assert fast == 0, "an all-ones significand must not take the fp32 path"
    assert fast_paths >= len(divisors) - 8


@pytest.mark.parametrize("model", ["mobilenet1.0", "mobilenet0.5"])
def test_a_net_with_recompute_pairs_equals_the_same_net_without(dev, ops, model):
    """quantize.fuse's recompute pairs (online input quantisation) change no value: logits, every block's current_input_max and
    the naive-EMA thresholds after a calibration step are bit-equal with FQ_RECOMPUTE on and off."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build as build
    rng = np.random.default_rng(11)
    X = mx.nd.array(rng.standard_normal((6, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0))
    outs = {}
    for on in (False, True):
        net = build(model, 1000, mx.gpu(0))
        net.fix_params()
        net.quantize_input(enable=True, online=True)
        net(mx.nd.NDArray(X._t[:2].contiguous()))
        fuse.fuse_inference(net)
        old = fuse.RECOMPUTE
        fuse.RECOMPUTE = on
        seen = []
        real = ops.pwdw_fused
        ops.pwdw_fused = lambda *a, **k: (seen.append(tuple(a[0].shape)), real(*a, **k))[1]
        try:
            out = net(X)
            cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
            net.update_ema()
            thr = np.asarray([b.input_max.data().asscalar() for b in net.collect_quantized_blocks()], np.float32)
        finally:
            fuse.RECOMPUTE = old
            ops.pwdw_fused = real
        outs[on] = (N(out._t), cur, thr, seen)
    assert len(outs[False][3]) == 0 and len(outs[True][3]) >= 2, outs[True][3]
    _eq(outs[True][0], outs[False][0], "logits")
    _eq(outs[True][1], outs[False][1], "current_input_max of every block")
    _eq(outs[True][2], outs[False][2], "thresholds after one naive-EMA step")


def test_a_block_called_on_its_own_returns_a_tensor_and_a_stray_deferred_array_is_materialised(dev, ops):
    """The deferred output of a recompute pair exists only inside the rewired net's forward: the 1x1 block called directly
    returns the stored tensor; and should a deferred array reach another consumer after all, that consumer computes it."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from quantization.mxnet_amd.quantize.convert import convert_conv2d as C
    from test_gpu_net import _build as build
    net = build("mobilenet1.0", 1000, mx.gpu(0))
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    X = mx.nd.array(np.random.default_rng(3).standard_normal((4, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0))
    net(X)
    fuse.fuse_inference(net)
    want = net(X).asnumpy()
    feats = list(net.features._children.values())
    pairs = [(i, b) for i, b in enumerate(feats) if getattr(b, "_fq_pw_fused", {}).get("pair_dw") is not None]
    assert len(pairs) >= 3
    # the layers in front of the first pair, then the 1x1 block on its own
    h = X
    i0, pw = pairs[0]
    for b in feats[:i0]:
        h = b(h)
    y = pw(h)
    assert y._fq_deferred is None and y._t.dtype == torch.float32 and tuple(y.shape) == (4, 64, 112, 112)
    # a deferred array handed to a block that is not its consumer
    seen = []
    real = C._materialise
    C._materialise = lambda a: (seen.append(1), real(a))[1]
    try:
        stat = ops.absmax_per_sample(h._t)
        codes = C._pointwise_weight_codes(pw, pw.quantize_args, pw.weight.data(), pw.weight.data())
        fz = pw._fq_pw_fused
        scale, shift = fz["constants"]()
        fake = mx.nd.NDArray(C._placeholder((4, 64, 112, 112), h._t.device))
        fake._fq_stat = ops.pwconv_i8_stat(h._t, *codes, None, in_stat=stat, bn_scale=scale, bn_shift=shift, act=fz["act"])
        fake._fq_deferred = dict(x=h._t, codes=codes, bias=None, bn=(scale, shift), act=fz["act"],
                                 plan=dict(in_stat=stat, width=8, flags=0, cur_out=torch.zeros(1, device=h._t.device)),
                                 consumer=object())
        z = fz["pair_dw"](fake)                                   # the record names another `consumer`: must materialise
        assert seen == [1] and z._t.dtype == torch.float32
    finally:
        C._materialise = real
    _eq(net(X).asnumpy(), want, "the net is unchanged by the detour")
