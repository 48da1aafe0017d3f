"""BASELINE.json configs 2-5 AS STATED and at FULL size on the MI355X: (128, 3, 224, 224) batches through the converted
nets with every quantised block checked while the net runs — the block's actual input and weight are copied to the host,
pushed through the C++/OpenMP oracle (oracle/libfq_host.so, pinned to the goldens in tests/test_host_oracle.py) and
must reproduce bit for bit what the HIP path handed to the convolution.  The numpy oracle needs minutes per layer at
these sizes; the C++ restatement of the same arithmetic needs seconds for a whole net.

  config 3  resnet50_v1, per-channel W8A8, offline KL calibration: disable_quantize -> collect_feature_maps ->
            kl_calibrate_many (53 layers, one launch) -> thresholds -> offline evaluation; every histogram and every
            best_bins against the oracle
  config 4  mobilenetv2_1.0, per-channel W4A8, naive-EMA calibration then offline evaluation
  config 5  resnet50_v1, Winograd-domain F43 per-channel weights, online
  config 2  mobilenet1.0 with the fused producers of the benchmark (stem / depthwise / pointwise-int8 / pooling kernels):
            every producer call at batch 128 against its host twin
"""
import os

import numpy as np
import pytest
import torch

from oracle import fq_oracle as O
from oracle import host as H

pytestmark = pytest.mark.gpu

FULL = (128, 3, 224, 224)


def _build(model, ctx, quant_type="layer", wt=8, in_w=8, signed=False, wino="none"):
    from test_gpu_net import _build as build
    return build(model, 1000, ctx, quant_type=quant_type, wt=wt, in_w=in_w, signed=signed, wino=wino)


def _images(ctx, seed, shape=FULL, scale=1.0):
    from quantization.mxnet_amd import mx
    g = torch.Generator(device=ctx.torch_device)
    g.manual_seed(seed)
    return mx.nd.NDArray(torch.randn(shape, device=ctx.torch_device, generator=g) * scale)


def _same(got, want, what):
    if not np.array_equal(got, want):
        bad = got != want
        raise AssertionError("%s: %d of %d elements differ (max |d| = %g)"
                             % (what, int(bad.sum()), bad.size, float(np.abs(got[bad] - want[bad]).max())))


class StreamingCheck(object):
    """Checks every quantised block against the host oracle DURING the forward (nothing is kept: a full-size ResNet-50
    forward moves 5 GB of block inputs)."""

    def __init__(self, net, signed=False, in_w=8, wt=8, quant_type="layer", wino="none"):
        self.args = dict(signed=signed, in_w=in_w, wt=wt, quant_type=quant_type, wino=wino)
        self.offline = False
        self.checked = 0
        self.elements = 0
        self.blocks = net.collect_quantized_blocks()
        for b in self.blocks:
            self._wrap(b)

    def _wrap(self, b):
        from quantization.mxnet_amd.mx.gluon import nn
        from quantization.mxnet_amd import ops
        orig, chk = b.origin_forward, self

        def pre(m, args):
            m._chk_raw = args[0]._t
            m._chk_fixed = getattr(m, "fixed_params", None)
            m._chk_w = m.weight.data()._t.detach().clone() if m._chk_fixed != 1 else None
        b.register_forward_pre_hook(pre)

        def wrapped(F, xq, wq, bias=None):
            a = chk.args
            x = b._chk_raw.detach().cpu().numpy()
            dense = isinstance(b, nn.Dense)
            flags = H.act_flags(a["signed"], lo_neg_max=False if dense else None)
            if chk.offline:
                thr = np.float32(b.input_max.data()._t.cpu().numpy()[0])
                want, cur, _ = H.fake_quant_offline(x, thr, a["in_w"], flags)
            else:
                want, cur, _ = H.fake_quant_online(x, a["in_w"], flags)
            _same(xq._t.detach().cpu().numpy(), want, "%s: fake-quantised input" % b.name)
            assert b._fq_cur.cpu().numpy()[0] == cur, "%s: current_input_max" % b.name
            if b._chk_w is None:
                assert wq._t.data_ptr() == b.weight.data()._t.data_ptr()      # frozen weights pass through (:96-97)
            else:
                w = b._chk_w.cpu().numpy()
                qt = a["quant_type"]
                if not dense and qt == "channel" and a["wino"] != "none" and tuple(w.shape[2:]) == (3, 3):
                    G, GI, GTI = ops.winograd_matrices(a["wino"])
                    want_w, _ = H.wino_weight_fake_quant(w, G, GI, GTI, a["wt"])
                else:
                    if dense:
                        rows = w.shape[0] if qt in ("channel", "group") else 1
                    else:
                        rows = {"layer": 1, "channel": w.shape[0], "group": b._kwargs["num_group"]}[qt]
                    want_w, _ = H.weight_fake_quant(w, rows, a["wt"])
                _same(wq._t.detach().cpu().numpy(), want_w, "%s: fake-quantised weight" % b.name)
            chk.checked += 1
            chk.elements += x.size
            return orig(F, xq, wq, bias)
        b.origin_forward = wrapped


def _property_checks(xq, scale_levels, signed):
    """size-independent properties of a fake-quantised tensor: integer codes within the range, y == codes * scale"""
    t = xq / scale_levels[0]
    codes = torch.round(t)
    assert torch.equal(codes * scale_levels[0], xq)
    lo = -scale_levels[1] if signed else 0
    assert float(codes.min()) >= lo and float(codes.max()) <= scale_levels[1]


# ---- config 3 -----------------------------------------------------------------------------------------------------------------
def test_config3_resnet50_per_channel_offline_kl_as_stated(gpu):
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps, kl_calibrate_many
    net = _build("resnet50_v1", gpu, quant_type="channel")
    blocks = net.collect_quantized_blocks()
    assert len(blocks) == 53
    # -- collect: two calibration batches of 128 (5.1 GB of block inputs each); the oracle histograms every block input as it goes by
    want_hist, want_max = {}, {}

    def watch(m, args):
        fm = args[0]._t.detach().cpu().numpy()
        if m not in want_max:
            want_max[m] = H.global_max(fm)                        # first batch fixes the range (:97-101)
            want_hist[m] = np.zeros(2048, np.uint64)
        H.histogram_accumulate(fm, want_max[m], 2048, want_hist[m])
    hooks = [b.register_forward_pre_hook(watch) for b in blocks]
    loader = [(_images(gpu, 100 + i, (128, 3, 224, 224)), None) for i in range(2)]     # the CLI's batch size (:81)
    net.disable_quantize()
    hists, maxes = collect_feature_maps(net, 2048, loader, gpu)
    for h in hooks:
        h.detach()
    for b in blocks:
        assert maxes[b] == want_max[b], b.name
        _same(hists[b], want_hist[b].astype(np.float32), "%s: histogram" % b.name)
        assert hists[b].sum() > 0
    # -- search: all 53 layers in one launch, every result against the oracle
    levels = 2 ** 8
    stacked = np.stack([hists[b] for b in blocks])
    best = kl_calibrate_many([hists[b] for b in blocks], levels, levels, 2048)
    _same(np.asarray(best, np.int32), H.kl_search(stacked, levels, levels), "best_bins of the 53 layers")
    assert best[0] == O.kl_calibrate(hists[blocks[0]], levels, levels, 2048)       # and the numpy oracle on one of them
    for b, k in zip(blocks, best):
        b.input_max.set_data(mx.nd.array([(k + 0.5) * (maxes[b] / 2048)], ctx=gpu))      # simulate_quantization.py:310
    # -- offline evaluation at the full batch, every block checked
    net.enable_quantize()
    net.fix_params()
    net.quantize_input(enable=True, online=False)
    chk = StreamingCheck(net, quant_type="channel")
    chk.offline = True
    X = _images(gpu, 7)
    out = net(X)
    assert chk.checked == 53 and chk.elements == 128 * 9987072          # 9 987 072 activation elements / image
    assert out.shape == (128, 1000) and bool(torch.isfinite(out._t).all())
    out2 = net(X)                                                        # frozen weights: same inputs, same thresholds
    assert chk.checked == 106
    # offline mode is batch-independent: the first 4 images alone give the same logits (up to MIOpen's solver choice)
    small = net(mx.nd.NDArray(X._t[:4].contiguous()))
    scale = float(out._t.abs().max())
    assert float((small._t - out._t[:4]).abs().max()) <= 2e-2 * scale
    assert float((out2._t - out._t).abs().max()) <= 2e-2 * scale


# ---- config 5 -----------------------------------------------------------------------------------------------------------------
def test_config5_resnet50_winograd_f43_online_full_batch(gpu):
    net = _build("resnet50_v1", gpu, quant_type="channel", wino="F43")
    chk = StreamingCheck(net, quant_type="channel", wino="F43")
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    out = net(_images(gpu, 8))
    assert chk.checked == 53 and bool(torch.isfinite(out._t).all())
    from quantization.mxnet_amd.mx.gluon import nn
    n3x3 = sum(1 for b in chk.blocks if isinstance(b, nn.Conv2D) and b._kwargs["kernel"] == (3, 3))
    assert n3x3 == 16                                                    # the Winograd-eligible convolutions


def test_config5_sliced_filters_change_the_logits_no_more_than_fusing_alone_full_batch(gpu):
    """The one place where the fused path multiplies something other than the reference's operand: under Winograd-domain weight
    quantisation the 3x3 layers take three int8 digit slices m p of the back-transformed filter g^ = GI U^ GTI
    (convert_conv2d.py:81-83; |g^ - m p| <= 2^-20 max|g^_c|, asserted per layer in test_gpu_net.py) instead of g^ itself
    (F.Convolution, :108).  Its effect on the logits at the BASELINE batch, measured against the two neighbours it can be told
    from (tools/sliced_effect.py): un-fused (A), fused with the fp32 filter through the tensor library (B), fused + sliced (C).
    Every pair differs by the last-bit noise that ANY other fp32 summation order has on this random-weight net (a few 8-bit
    rounding decisions flip downstream) - the sliced pair B-C no more than fusing alone, A-B - and the top-1 class agrees on
    every image."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import sliced_effect
    scale, out, n = sliced_effect.measure(128, "resnet50_v1", "F43")
    assert n == 128 and scale > 0
    for pair, (mx_, mean_, agree) in out.items():
        assert mx_ <= 2e-2 * scale, (pair, mx_, scale)            # measured: 0.8-0.9 % of max|logit| for all three pairs
        assert agree == n, (pair, agree)
    assert out["B-C"][1] <= 1.5 * out["A-B"][1], out              # the slices add nothing beyond the noise fusing alone has


# ---- config 4 -----------------------------------------------------------------------------------------------------------------
def test_config4_mobilenetv2_w4_naive_ema_then_offline_full_batch(gpu):
    net = _build("mobilenetv2_1.0", gpu, quant_type="channel", wt=4)
    chk = StreamingCheck(net, quant_type="channel", wt=4)
    blocks = chk.blocks
    state = np.zeros(len(blocks), np.float32)
    net.quantize_input(enable=True, online=True)
    for step in range(2):                                                # calibration steps at the full batch
        net(_images(gpu, 20 + step, scale=1.0 + 0.5 * step))
        net.update_ema()
        cur = np.asarray([float(b.current_input_max) for b in blocks], np.float32)
        state = O.ema_update(state, cur, 0.9)
        got = np.asarray([b.input_max.data().asscalar() for b in blocks], np.float32)
        _same(got, state, "EMA step %d" % step)
    assert chk.checked == 2 * len(blocks)
    net.fix_params()
    net.quantize_input(enable=True, online=False)
    chk.offline = True
    out = net(_images(gpu, 30))
    assert chk.checked == 3 * len(blocks) and bool(torch.isfinite(out._t).all())
    # linear bottlenecks hand NEGATIVE values to the expansion convolutions; with unsigned inputs they clip to 0 (kept)
    assert any(float(b._chk_raw.min()) < 0 for b in blocks)


# ---- config 2 with the benchmark's fused producers ------------------------------------------------------------------------------
def test_config2_mobilenet_fused_producers_full_batch_against_host_twins(gpu):
    """The benchmark's step: every call of the stem / depthwise / pointwise-int8 / pooling producers at batch 128 is
    recomputed by its C++ twin from the call's actual input."""
    from quantization.mxnet_amd import ops
    from quantization.mxnet_amd.quantize import fuse
    net = _build("mobilenet1.0", gpu)
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    X = _images(gpu, 9)
    net(mx_small(X))                                                     # freeze the weights on a small batch first
    assert fuse.fuse_inference(net, dense_int8=True) > 0           # (the classifier on the integer codes as well)
    seen = {"dw": 0, "pw": 0, "stem": 0, "gap": 0}
    real = {k: getattr(ops, k) for k in ("dwconv3x3", "pwconv_i8", "stem_conv_s2", "global_avg_pool_stat")}

    def np_(t):
        return None if t is None else t.detach().cpu().numpy()

    def dw(x, w, bias=None, **k):
        y, stat = real["dwconv3x3"](x, w, bias, **k)
        want, wstat = H.dwconv3x3(np_(x), np_(w), np_(bias), k["stride"], in_stat=np_(k["in_stat"]),
                                  bn_scale=np_(k["bn_scale"]), bn_shift=np_(k["bn_shift"]), act=k["act"], want_stat=True)
        _same(np_(k["in_stat"]), H.absmax_per_sample(np_(x)), "depthwise: statistic handed over by the producer")
        _same(np_(y), want, "depthwise 3x3 %s" % (tuple(x.shape),))
        _same(np_(stat), wstat, "depthwise statistic")
        seen["dw"] += 1
        return y, stat

    def pw(x, codes, scales, rowsum, bias=None, **k):
        y, stat = real["pwconv_i8"](x, codes, scales, rowsum, bias, **k)
        cout, cin = y.shape[1], x.shape[1]
        cin_pad = codes.shape[1]                                        # [rows_pad][cin_pad] int8, zero padded
        wc = np.ascontiguousarray(np_(codes))
        want = np.empty(tuple(y.shape), np.float32)
        wstat = np.zeros(x.shape[0], np.float32)
        cur = np.empty(1, np.float32)
        H._call("fq_pwconv_i8_host", np_(x), wc, np_(scales), np_(rowsum), np_(bias), want, x.shape[0], cin, cin_pad, cout,
                x.shape[2] * x.shape[3], np_(k["in_stat"]), None, H._i(k["width"]), H._u(k["flags"]), cur,
                np_(k.get("bn_scale")), np_(k.get("bn_shift")), H._i(H._ACTS[k.get("act")]), wstat, None, None)
        _same(np_(y), want, "pointwise int8 %s -> %d" % (tuple(x.shape), cout))
        if k.get("want_stat", True):
            _same(np_(stat), wstat, "pointwise statistic")
        else:
            assert stat is None                                         # the classifier (Dense on a 1x1 plane): logits only
        assert np_(k["cur_out"])[0] == cur[0]
        seen["pw"] += 1
        return y, stat

    def stem(x, w, bias=None, **k):
        y, stat = real["stem_conv_s2"](x, w, bias, **k)
        want, wstat = H.stem_conv_s2(np_(x), np_(w), np_(bias), np_(k["bn_scale"]), np_(k["bn_shift"]), k["act"],
                                       want_stat=True)
        _same(np_(y), want, "stem convolution")
        _same(np_(stat), wstat, "stem statistic")
        seen["stem"] += 1
        return y, stat

    def gap(x, **k):
        y, stat = real["global_avg_pool_stat"](x, **k)
        want, wstat = H.global_avg_pool(np_(x), want_stat=True)
        _same(np_(y).reshape(want.shape), want, "global average pool")
        _same(np_(stat), wstat, "pooling statistic")
        seen["gap"] += 1
        return y, stat
    # the recompute pairs (round 6): the statistic-only pass against the twin of the storing kernel, and the fused launch
    # against the twins of the two launches it replaces, fed with the launch's own inputs
    real["pwconv_i8_stat"], real["pwdw_fused"] = ops.pwconv_i8_stat, ops.pwdw_fused

    def pw_host(x, codes, scales, rowsum, bias, in_stat, width, flags, bn_scale, bn_shift, act):
        cout, cin = scales.numel(), x.shape[1]
        want = np.empty((x.shape[0], cout, x.shape[2], x.shape[3]), np.float32)
        wstat = np.zeros(x.shape[0], np.float32)
        cur = np.empty(1, np.float32)
        H._call("fq_pwconv_i8_host", np_(x), np.ascontiguousarray(np_(codes)), np_(scales), np_(rowsum), np_(bias), want,
                x.shape[0], cin, codes.shape[1], cout, x.shape[2] * x.shape[3], np_(in_stat), None, H._i(width), H._u(flags), cur,
                np_(bn_scale), np_(bn_shift), H._i(H._ACTS[act]), wstat, None, None)
        return want, wstat, cur

    def pw_stat(x, codes, scales, rowsum, bias=None, **k):
        stat = real["pwconv_i8_stat"](x, codes, scales, rowsum, bias, **k)
        _, wstat, cur = pw_host(x, codes, scales, rowsum, bias, k["in_stat"], k["width"], k["flags"], k.get("bn_scale"),
                                k.get("bn_shift"), k.get("act"))
        _same(np_(stat), wstat, "statistic-only pointwise pass %s" % (tuple(x.shape),))
        assert np_(k["cur_out"])[0] == cur[0]
        seen["pw"] += 1
        seen["pairs"] = seen.get("pairs", 0) + 1
        return stat

    def pwdw(x, codes, scales, rowsum, dw_w, **k):
        z, zstat = real["pwdw_fused"](x, codes, scales, rowsum, dw_w, **k)
        y, ystat, _ = pw_host(x, codes, scales, rowsum, k.get("pw_bias"), k["in_stat"], k["width"], k["flags"],
                              k.get("pw_bn_scale"), k.get("pw_bn_shift"), k.get("pw_act"))
        _same(np_(k["mid_stat"]), ystat, "statistic handed to the recomputing launch")
        want, wstat = H.dwconv3x3(y, np_(dw_w), np_(k.get("dw_bias")), k["stride"], in_stat=ystat, bn_scale=np_(k["dw_bn_scale"]),
                                  bn_shift=np_(k["dw_bn_shift"]), act=k["dw_act"], want_stat=True)
        _same(np_(z), want, "recomputing pointwise + depthwise launch %s" % (tuple(x.shape),))
        _same(np_(zstat), wstat, "its statistic")
        seen["dw"] += 1
        return z, zstat
    # the pooled producer (round 6): the last 1x1 and the pooling behind it in one launch, against the twins of both
    real["pwconv_i8_gap"] = ops.pwconv_i8_gap

    def pw_gap(x, codes, scales, rowsum, bias=None, **k):
        y, stat = real["pwconv_i8_gap"](x, codes, scales, rowsum, bias, **k)
        full, _, cur = pw_host(x, codes, scales, rowsum, bias, k["in_stat"], k["width"], k["flags"], k.get("bn_scale"),
                               k.get("bn_shift"), k.get("act"))
        want, wstat = H.global_avg_pool(full, want_stat=True)
        _same(np_(y).reshape(want.shape), want, "pointwise + global average pool in one launch %s" % (tuple(x.shape),))
        _same(np_(stat), wstat, "its statistic")
        assert np_(k["cur_out"])[0] == cur[0]
        seen["pw"] += 1
        seen["gap"] += 1
        return y, stat
    ops.dwconv3x3, ops.pwconv_i8, ops.stem_conv_s2, ops.global_avg_pool_stat = dw, pw, stem, gap
    ops.pwconv_i8_stat, ops.pwdw_fused, ops.pwconv_i8_gap = pw_stat, pwdw, pw_gap
    try:
        out = net(X)
    finally:
        for k, fn in real.items():
            setattr(ops, k, fn)
    pairs = seen.pop("pairs", 0)
    assert seen == {"dw": 13, "pw": 14, "stem": 1, "gap": 1}, seen          # 13 pointwise layers + the classifier
    assert pairs >= (2 if fuse.RECOMPUTE else 0), pairs                     # 32->64 @112 and 64->128 @56 at least
    assert bool(torch.isfinite(out._t).all())


def mx_small(X):
    from quantization.mxnet_amd import mx
    return mx.nd.NDArray(X._t[:2].contiguous())


# ---- the ADVICE items that need a device ------------------------------------------------------------------------------------------
def test_fused_caches_follow_in_place_parameter_updates(gpu):
    """fuse -> eval -> set_data with new weights / BatchNorm statistics -> eval must equal a freshly fused net: the folded
    BatchNorm constants and the frozen pointwise weight codes are keyed on the parameters' storage AND version."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build as build
    rng = np.random.default_rng(3)
    X = mx.nd.array(rng.standard_normal((4, 3, 64, 64)).astype(np.float32), ctx=gpu)

    def fresh():
        net = build("mobilenet1.0", 1000, gpu)
        net.fix_params()
        net.quantize_input(enable=True, online=True)
        return net

    def perturb(net):
        r = np.random.default_rng(17)
        for b in net.collect_quantized_blocks():
            if isinstance(b, nn.Conv2D) and b._kwargs["kernel"] == (1, 1):
                shape = b.weight.shape                # NEW values (the current ones differ: frozen vs raw)
                b.weight.set_data(mx.nd.array((r.standard_normal(shape) * 0.05).astype(np.float32), ctx=gpu))

        def bn(m):
            if type(m) is nn.BatchNorm:
                shape = m.running_var.shape
                m.running_var.set_data(mx.nd.array(r.uniform(0.5, 2.0, shape).astype(np.float32), ctx=gpu))
        net.apply(bn)
    a = fresh()
    a(X)
    fuse.fuse_inference(a)
    a(X)                                        # caches are warm now
    perturb(a)
    a.fix_params()                              # re-freeze the new weights
    got = a(X).asnumpy()
    b = fresh()
    perturb(b)
    b.fix_params()
    b(X)
    fuse.fuse_inference(b)
    want = b(X).asnumpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-2 * np.abs(want).max())
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() or np.array_equal(got, want)


def test_entry_points_follow_the_tensor_device(gpu):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices (the pool's boxes have one)")
    from quantization.mxnet_amd import ops
    x = torch.randn(8, 16, 14, 14, device="cuda:1")
    y, cur, _ = ops.fake_quant_online(x, 8, 0)
    want, wcur, _ = H.fake_quant_online(x.cpu().numpy())
    _same(y.cpu().numpy(), want, "fake-quant on device 1")
    assert cur.cpu().numpy()[0] == wcur
