"""GPU whole-net parity for the BASELINE.json configs (reduced spatial size so the oracle finishes in seconds):
every quantised block of the converted net is spied on while the net runs on the MI355X; each block's RAW input and
weight are then pushed through the CPU oracle and must reproduce, bit for bit, what the HIP path handed to the
convolution.  (End-to-end logits cannot be compared bit-exactly across devices: MIOpen and the CPU convolution sum in
different orders, so activations differ in the last bits BEFORE they reach the next fake-quant.)"""
import contextlib

import numpy as np
import pytest
import torch

from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu


def _build(model, classes, ctx, quant_type="layer", wt=8, in_w=8, signed=False, wino="none"):
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.mx.gluon.model_zoo import get_model
    from quantization.mxnet_amd.quantize import convert
    from quantization.mxnet_amd.quantize.initialize import qparams_init
    np.random.seed(7)
    net = get_model(model, classes=classes)
    convert_fn = {nn.Conv2D: convert.gen_conv2d_converter(quantize_input=True, wino_quantize=wino, input_signed=signed,
                                                          weight_width=wt, input_width=in_w, quant_type=quant_type),
                  nn.Dense: convert.gen_dense_converter(quantize_input=True, input_signed=signed, weight_width=wt,
                                                        input_width=in_w, quant_type=quant_type),
                  nn.Activation: None, nn.BatchNorm: None}
    exclude = [net.features[0], net.features[1]]
    if model.startswith("mobilenetv2_"):
        exclude.append(net.output[0])
    if model.startswith("cifar_resnet"):
        exclude.extend([net.features[2][0].body[0], net.features[2][0].body[1]])
    convert.convert_model(net, exclude=exclude, convert_fn=convert_fn)
    qparams_init(net)
    net.collect_params().reset_ctx(ctx)
    if model.startswith("vgg"):
        # (the first Dense layer's input width follows from the image size - deferred initialisation, as in gluon; the Spy's
        # pre-hooks read the weights before the block's own forward would materialise them)
        net._fq_test_warm = True
    return net


class Spy(object):
    """Capture, per quantised block and per forward: raw input, raw weight, what reached origin_forward."""

    def __init__(self, net):
        self.records = []
        self.blocks = net.collect_quantized_blocks()
        for b in self.blocks:
            self._wrap(b)

    def _wrap(self, b):
        orig = b.origin_forward
        spy = self

        def pre(m, args):
            m._spy_raw = args[0]._t.detach().clone()
            m._spy_w = m.weight.data()._t.detach().clone()
            m._spy_fixed = getattr(m, "fixed_params", None)
        b.register_forward_pre_hook(pre)

        def wrapped(F, xq, wq, bias=None):
            spy.records.append(dict(block=b, x=b._spy_raw, w=b._spy_w, xq=xq._t.detach().clone(),
                                    wq=wq._t.detach().clone(), fixed=b._spy_fixed,
                                    cur=b._fq_cur.detach().clone() if hasattr(b, "_fq_cur") else None,
                                    thr=b.input_max.data()._t.detach().clone()))
            return orig(F, xq, wq, bias)
        b.origin_forward = wrapped


def _check_records(records, signed, in_w, wt, quant_type, wino, offline, allow_empty=False):
    from quantization.mxnet_amd.mx.gluon import nn
    assert records or allow_empty
    for r in records:
        b = r["block"]
        x, xq = r["x"].cpu().numpy(), r["xq"].cpu().numpy()
        thr = np.float32(r["thr"].cpu().numpy()[0]) if offline else None
        if isinstance(b, nn.Dense):
            want, cur, _, _ = O.dense_input_fake_quant(x, signed, in_w, offline_threshold=thr)
        else:
            want, cur, _, _ = O.conv_input_fake_quant(x, signed, in_w, offline_threshold=thr)
        assert np.array_equal(xq, want), "%s: activation fake-quant differs from the oracle" % b.name
        assert r["cur"].cpu().numpy()[0] == cur, "%s: current_input_max" % b.name
        w, wq = r["w"].cpu().numpy(), r["wq"].cpu().numpy()
        if isinstance(b, nn.Conv2D) and r["fixed"] == 1:
            assert np.array_equal(wq, w)                      # frozen weights pass through (convert_conv2d.py:96-97)
            continue
        if isinstance(b, nn.Conv2D) and quant_type == "channel" and wino != "none" and tuple(w.shape[2:]) == (3, 3):
            want_w, _, _ = O.wino_weight_fake_quant(w, wino, wt)
        elif isinstance(b, nn.Dense):
            want_w, _ = O.weight_fake_quant(w, "channel" if quant_type in ("channel", "group") else "layer", wt)
        else:
            want_w, _ = O.weight_fake_quant(w, quant_type, wt, num_group=b._kwargs["num_group"])
        assert np.array_equal(wq, want_w), "%s: weight fake-quant differs from the oracle" % b.name


CONFIGS = [
    # (BASELINE config, model, classes, hw, batch, kwargs)
    ("cfg1 cifar_resnet20_v1 per-layer W8A8 online", "cifar_resnet20_v1", 10, 32, 8, dict()),
    ("cfg2 mobilenet1.0 per-layer W8A8 online", "mobilenet1.0", 1000, 64, 4, dict()),
    ("cfg3 resnet50_v1 per-channel W8A8", "resnet50_v1", 1000, 64, 2, dict(quant_type="channel")),
    ("cfg4 mobilenetv2_1.0 per-channel W4A8", "mobilenetv2_1.0", 1000, 64, 4, dict(quant_type="channel", wt=4)),
    ("cfg5 resnet50_v1 Winograd F43 per-channel", "resnet50_v1", 1000, 64, 2, dict(quant_type="channel", wino="F43")),
    ("signed int8 inputs, group-wise weights", "mobilenet1.0", 1000, 32, 4, dict(signed=True, quant_type="group")),
    # (the net of the reference's own tests/test_collect_qparams.py: 3x3 convolutions with bias and without BatchNorm, 2x2 pooling,
    # Dense layers with an activation of their own and an unflattened input)
    ("vgg vgg11 per-channel W8A8 online", "vgg11", 10, 32, 4, dict(quant_type="channel")),
]


@pytest.mark.parametrize("name,model,classes,hw,batch,kw", CONFIGS, ids=[c[0].split()[0] + "_" + c[1] for c in CONFIGS])
def test_every_quantised_block_matches_oracle_online_then_frozen(gpu, name, model, classes, hw, batch, kw):
    from quantization.mxnet_amd import mx
    net = _build(model, classes, gpu, **kw)
    rng = np.random.default_rng(7)
    X = mx.nd.array(rng.standard_normal((batch, 3, hw, hw)).astype(np.float32), ctx=gpu)
    if getattr(net, "_fq_test_warm", False):
        net(X)
    spy = Spy(net)
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    out = net(X)
    assert out.shape == (batch, classes) and np.isfinite(out.asnumpy()).all()
    args = dict(signed=kw.get("signed", False), in_w=kw.get("in_w", 8), wt=kw.get("wt", 8),
                quant_type=kw.get("quant_type", "layer"), wino=kw.get("wino", "none"))
    _check_records(spy.records, offline=False, **args)
    n_first = len(spy.records)
    assert n_first == len(spy.blocks)
    from quantization.mxnet_amd.mx.gluon import nn
    assert all(b.fixed_params == 1 for b in spy.blocks if isinstance(b, nn.Conv2D))
    spy.records.clear()
    net(X)                                                    # second forward: weights frozen
    _check_records(spy.records, offline=False, **args)


def test_naive_ema_calibration_then_offline_eval(gpu):
    """Config 4's flow (simulate_quantization.py:320-323,337-339) on the GPU; EMA checked step by step."""
    from quantization.mxnet_amd import mx
    net = _build("mobilenetv2_1.0", 1000, gpu, quant_type="channel", wt=4)
    spy = Spy(net)
    blocks = spy.blocks
    rng = np.random.default_rng(11)
    state = np.zeros(len(blocks), np.float32)
    net.quantize_input(enable=True, online=True)
    for step in range(4):
        X = mx.nd.array(rng.standard_normal((4, 3, 64, 64)).astype(np.float32) * (1 + 0.3 * step), ctx=gpu)
        spy.records.clear()
        net(X)
        net.update_ema()
        cur = np.asarray([float(b.current_input_max) for b in blocks], np.float32)
        state = O.ema_update(state, cur, 0.9)
        got = np.asarray([b.input_max.data().asscalar() for b in blocks], np.float32)
        assert np.array_equal(got, state), "EMA step %d" % step
        _check_records(spy.records, signed=False, in_w=8, wt=4, quant_type="channel", wino="none", offline=False)
    arena = net.calibration_arena()
    assert arena.state.is_cuda and arena.state.numel() == len(blocks)
    net.fix_params()
    net.quantize_input(enable=True, online=False)
    spy.records.clear()
    net(X)
    _check_records(spy.records, signed=False, in_w=8, wt=4, quant_type="channel", wino="none", offline=True)


def test_kl_calibration_flow(gpu):
    """Config 3's flow (simulate_quantization.py:294-315): disable_quantize -> collect_feature_maps -> kl_calibrate ->
    thresholds -> offline eval; histograms and best_bins against the oracle on the captured block inputs."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps, kl_calibrate, \
        kl_calibrate_many
    net = _build("cifar_resnet20_v1", 10, gpu, quant_type="channel")
    blocks = net.collect_quantized_blocks()
    captured = {b: [] for b in blocks}
    hooks = [b.register_forward_pre_hook(lambda m, a: captured[m].append(a[0].asnumpy())) for b in blocks]
    rng = np.random.default_rng(5)
    loader = [(mx.nd.array(rng.standard_normal((8, 3, 32, 32)).astype(np.float32)), None) for _ in range(3)]
    net.disable_quantize()
    hists, maxes = collect_feature_maps(net, 2048, loader, gpu)
    for h in hooks:
        h.detach()
    assert set(hists) == set(blocks)
    for b in blocks:
        want, fm_max = None, None
        for fm in captured[b]:
            h, m = O.discrete_histogram(fm, 2048, fm_max)
            fm_max = m if fm_max is None else fm_max
            want = h if want is None else want + h
        assert maxes[b] == fm_max and np.array_equal(hists[b], want), b.name
    levels = 2 ** 8
    best_all = kl_calibrate_many([hists[b] for b in blocks], levels, levels, 2048)
    for b, best in list(zip(blocks, best_all))[:3] + list(zip(blocks, best_all))[-2:]:
        assert best == O.kl_calibrate(hists[b], levels, levels, 2048), b.name
        assert kl_calibrate(hists[b], levels=levels, min_bins=levels, bins=2048) == best
    for b, best in zip(blocks, best_all):
        th = (best + 0.5) * (maxes[b] / 2048)
        b.input_max.set_data(mx.nd.array([th], ctx=gpu))
    net.enable_quantize()
    net.fix_params()
    net.quantize_input(enable=True, online=False)
    spy = Spy(net)
    out = net(loader[0][0].as_in_context(gpu))
    assert np.isfinite(out.asnumpy()).all()
    _check_records(spy.records, signed=False, in_w=8, wt=8, quant_type="channel", wino="none", offline=True)


def test_nn_conv2d_int_code_path_on_gpu(gpu, golden):
    from quantization.mxnet_amd import mx, nn as qnn
    g = golden("g8_quantized_conv")
    for use_bias in (0, 1):
        for groups in (1, 2):
            tag = "conv_b%d_g%d" % (use_bias, groups)
            c = qnn.Conv2D(10, 3, 1, 1, in_channels=2, groups=groups, use_bias=bool(use_bias), quantized=True,
                           input_dtype="uint8", weight_dtype="int8")
            c.initialize(ctx=gpu)
            c.weight.set_data(mx.nd.array(g[tag + "/w"]))
            if use_bias:
                c.bias.set_data(mx.nd.array(g[tag + "/b"]))
            y = c(mx.nd.array(g[tag + "/x"], ctx=gpu)).asnumpy()
            assert np.array_equal(y, g[tag + "/y_int"]), tag
            assert np.abs(y - g[tag + "/y_sim"]).max() < 0.1 and np.abs(y - g[tag + "/y_float"]).max() < 0.1


@pytest.mark.parametrize("model,classes,hw,batch,kw", [("mobilenet1.0", 1000, 64, 4, dict()),
                                                       ("mobilenetv2_1.0", 1000, 64, 4, dict(quant_type="channel", wt=4)),
                                                       ("resnet50_v1", 1000, 64, 2, dict(quant_type="channel")),
                                                       ("resnet50_v1", 1000, 64, 2, dict(quant_type="channel", wino="F43")),
                                                       ("cifar_resnet20_v1", 10, 32, 8, dict()),
                                                       ("vgg11", 10, 32, 4, dict(quant_type="channel")),
                                                       ("vgg11_bn", 10, 64, 2, dict())],
                         ids=["mobilenet1.0", "mobilenetv2_1.0", "resnet50_v1", "resnet50_v1-wino-F43", "cifar_resnet20_v1", "vgg11", "vgg11_bn"])
def test_fused_inference_keeps_every_block_exact(gpu, model, classes, hw, batch, kw):
    """quantize/fuse.py: BN+ReLU(+statistic) in one pass, consumer skips its statistic pass.  Every quantised block
    must still be exactly oracle(its actual input); logits stay within BN-formula rounding of the unfused net."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    net = _build(model, classes, gpu, **kw)
    rng = np.random.default_rng(7)
    X = mx.nd.array(rng.standard_normal((batch, 3, hw, hw)).astype(np.float32), ctx=gpu)
    if getattr(net, "_fq_test_warm", False):
        net(X)
    raw3x3 = {id(b): b.weight.data()._t.detach().cpu().numpy().copy() for b in net.collect_quantized_blocks()
              if getattr(b, "_kwargs", {}).get("kernel") == (3, 3)}
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    ref = net(X).asnumpy()
    n = fuse.fuse_inference(net, dense_int8=(model != "mobilenet1.0"))      # the classifier on the codes too, and off
    assert n > 0
    spy = Spy(net)
    # depthwise convolutions taken over by fq_dwconv3x3 never reach origin_forward: record the kernel calls instead
    from quantization.mxnet_amd import ops
    dw_calls = []
    real_dw = ops.dwconv3x3

    def spy_dw(x, w, bias=None, **k):
        y, stat = real_dw(x, w, bias, **k)
        dw_calls.append(dict(x=x.detach().clone(), w=w.detach().clone(), y=y.detach().clone(), stat=stat.detach().clone(),
                             k={a: (b.detach().clone() if torch.is_tensor(b) else b) for a, b in k.items()}))
        return y, stat
    pw_calls = []
    real_pw = ops.pwconv_i8

    def spy_pw(x, codes, scales, rowsum, bias=None, **k):
        y, stat = real_pw(x, codes, scales, rowsum, bias, **k)
        pw_calls.append(dict(x=x.detach().clone(), codes=codes.detach().clone(), scales=scales.detach().clone(),
                             rowsum=rowsum.detach().clone(), bias=None if bias is None else bias.detach().clone(),
                             y=y.detach().clone(), stat=None if stat is None else stat.detach().clone(),
                             k={a: (b.detach().clone() if torch.is_tensor(b) else b) for a, b in k.items()}))
        return y, stat
    c3_calls = []
    real_c3 = ops.conv3x3_i8

    def spy_c3(x, codes, scales, rowsum, bias=None, **k):
        y, stat = real_c3(x, codes, scales, rowsum, bias, **k)
        c3_calls.append(dict(x=x.detach().clone(), codes=codes.detach().clone(), scales=scales.detach().clone(),
                             rowsum=rowsum.detach().clone(), bias=None if bias is None else bias.detach().clone(),
                             y=y.detach().clone(), stat=stat.detach().clone(),
                             k={a: (b.detach().clone() if torch.is_tensor(b) else b) for a, b in k.items()}))
        return y, stat
    ops.dwconv3x3 = spy_dw
    ops.pwconv_i8 = spy_pw
    ops.conv3x3_i8 = spy_c3
    try:
        out = net(X).asnumpy()
    finally:
        ops.dwconv3x3 = real_dw
        ops.pwconv_i8 = real_pw
        ops.conv3x3_i8 = real_c3
    args = dict(signed=False, in_w=8, wt=kw.get("wt", 8), quant_type=kw.get("quant_type", "layer"), wino=kw.get("wino", "none"))
    _check_records(spy.records, offline=False, allow_empty=True, **args)   # (mobilenetv2: every block is taken over)
    n_dw = sum(1 for b in spy.blocks if hasattr(b, "_fq_dw_fused"))
    # (a quantised Dense is a 1x1 convolution on a 1x1 plane: it goes through fq_pwconv_i8 as well)
    # (... unless it has an activation of its own - vgg's Dense(4096, relu) - which keeps it with the library's GEMM)
    n_pw = sum(1 for b in spy.blocks if hasattr(b, "_fq_pw_fused") or
               (getattr(b, "_fq_dense_int8", False) and getattr(b, "act", None) is None))
    assert len(dw_calls) == n_dw and len(pw_calls) + len(c3_calls) == n_pw
    assert len(spy.records) + n_dw + n_pw == len(spy.blocks)
    # the 3x3 layers with 64 ... 512 input channels run on the integer codes: the ResNet-50 bottlenecks, the last stage of
    # the CIFAR ResNet-20
    assert (len(c3_calls) > 0) == (model in ("resnet50_v1", "cifar_resnet20_v1") or model.startswith("vgg"))
    from oracle import patch as OP
    if kw.get("wino", "none") != "none":
        # config 5: the 3x3 layers of the bottlenecks leave MIOpen - the filter they multiply is the oracle's Winograd-domain
        # fake-quantised filter (convert_conv2d.py:71-83), cut into three int8 digit slices whose value m * p reproduces it to
        # p / 2 <= 2^-20 of the channel maximum
        assert len(c3_calls) == 16 and all(c["codes"].dim() == 2 and c["codes"].shape[0] == 3 for c in c3_calls)
        wino_blocks = [b for b in spy.blocks if id(b) in raw3x3 and getattr(b, "_fq_pw_fused", {}).get("sliced")]
        assert len(wino_blocks) == 16
        for b, call in zip(wino_blocks, c3_calls):
            want_w = O.wino_weight_fake_quant(raw3x3[id(b)], kw["wino"], 8)[0]
            assert np.array_equal(b.weight.data()._t.cpu().numpy(), want_w), "frozen Winograd-domain filter"
            cout, cin = want_w.shape[:2]
            rows_pad = (cout + 63) // 64 * 64
            codes = call["codes"].cpu().numpy().astype(np.int64)
            d = [codes[sl, :rows_pad * 9 * cin].reshape(rows_pad, 9 * cin)[:cout] for sl in range(3)]
            m = ((d[0] << 14) + (d[1] << 7) + d[2]).reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
            p = call["scales"].cpu().numpy().astype(np.float64)
            err = np.abs(m * p[:, None, None, None] - want_w.astype(np.float64)).reshape(cout, -1).max(axis=1)
            assert np.all(err <= 2.0 ** -20 * np.abs(want_w).reshape(cout, -1).max(axis=1) + 1e-30)
    for call in c3_calls:
        k = call["k"]
        x_raw = call["x"].cpu()
        per_sample = O.absmax_per_sample(x_raw.numpy())
        assert np.array_equal(k["in_stat"].cpu().numpy(), per_sample)
        assert k["cur_out"].cpu().numpy()[0] == O.batch_mean(per_sample)
        cpu_k = {a: (b.cpu() if torch.is_tensor(b) else b) for a, b in k.items()}
        cpu_k["cur_out"] = torch.zeros(1)
        want, want_stat = OP.conv3x3_i8(x_raw, call["codes"].cpu(), call["scales"].cpu(), call["rowsum"].cpu(),
                                        None if call["bias"] is None else call["bias"].cpu(), **cpu_k)
        assert np.array_equal(call["y"].cpu().numpy(), want.numpy()), "3x3 int8 conv differs from the oracle"
        assert np.array_equal(call["stat"].cpu().numpy(), want_stat.numpy())
    for call in pw_calls:
        k = call["k"]
        x_raw = call["x"].cpu()
        per_sample = O.absmax_per_sample(x_raw.numpy())
        assert np.array_equal(k["in_stat"].cpu().numpy(), per_sample)
        assert k["cur_out"].cpu().numpy()[0] == O.batch_mean(per_sample)
        cpu_k = {a: (b.cpu() if torch.is_tensor(b) else b) for a, b in k.items()}
        cpu_k["cur_out"] = torch.zeros(1)
        want, want_stat = OP.pwconv_i8(x_raw, call["codes"].cpu(), call["scales"].cpu(), call["rowsum"].cpu(),
                                       None if call["bias"] is None else call["bias"].cpu(), **cpu_k)
        assert np.array_equal(call["y"].cpu().numpy(), want.numpy()), "pointwise int8 conv differs from the oracle"
        assert (call["stat"] is None) == (want_stat is None)
        if want_stat is not None:
            assert np.array_equal(call["stat"].cpu().numpy(), want_stat.numpy())
    for call in dw_calls:
        k = call["k"]
        x_raw = call["x"].cpu().numpy()
        per_sample = O.absmax_per_sample(x_raw)
        assert np.array_equal(k["in_stat"].cpu().numpy(), per_sample)           # hint == what the statistic pass gives
        assert k["cur_out"].cpu().numpy()[0] == O.batch_mean(per_sample)
        want = O.dwconv3x3(x_raw, call["w"].cpu().numpy(), None, k["stride"], O.batch_mean(per_sample), False, 8, None,
                           k["bn_scale"].cpu().numpy(), k["bn_shift"].cpu().numpy(), k["act"])
        got = call["y"].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
        assert np.array_equal(call["stat"].cpu().numpy(), O.absmax_per_sample(got))
    # end to end this is only a sanity bound: the fused producers (own stem / depthwise / pointwise kernels, folded BN)
    # differ from MIOpen in the last bit, which flips a few rounding decisions of the following 8-bit quantisers on this
    # random-weight net; exactness is established block by block above
    scale = np.abs(ref).max()
    assert np.abs(out - ref).max() <= 5e-2 * scale, (np.abs(out - ref).max(), scale)
    fuse.unfuse(net)
    again = net(X).asnumpy()
    # (not bit-equal across calls: MIOpen may pick a different convolution solver once its find-db is warm)
    assert np.abs(again - ref).max() <= 2e-2 * scale


def test_fake_bn_flow_on_gpu_matches_reference_goldens(gpu, golden):
    """--merge-bn (fake-BN fold + bypass_bn) on the device, against G10 made by the reference's own code."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_logic import run_fake_bn_flow
    g = golden("g10_fake_bn")
    out = run_fake_bn_flow(g, ctx=gpu)
    for key, ref in (("calib", "fakebn/calib_logits"), ("frozen0", "fakebn/frozen_logits0"),
                     ("frozen1", "fakebn/frozen_logits1")):
        np.testing.assert_allclose(out[key], g[ref], rtol=2e-3, atol=2e-3)       # MIOpen vs CPU convolution order
    for name, v in out["params"].items():
        if name.endswith(("weight", "bias")) and "conv0" not in name and "dense" not in name:
            np.testing.assert_allclose(v, g["fakebn/frozen/" + name], rtol=1e-5, atol=1e-6, err_msg=name)


def test_dense_weight_cache_follows_the_parameter(gpu):
    """convert_dense keeps the fake-quantised weight between forwards (the reference recomputes it, :52-63) — it must be
    refreshed as soon as the parameter changes in place or is replaced."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.quantize import convert
    net = nn.HybridSequential()
    net.add(nn.Dense(8, in_units=16))
    net.initialize(ctx=gpu)
    convert.convert_model(net, convert_fn={nn.Dense: convert.gen_dense_converter(quantize_input=False)})
    dense = net[0]
    rng = np.random.default_rng(5)
    X = mx.nd.array(rng.standard_normal((4, 16)).astype(np.float32), ctx=gpu)

    def expect():
        w = dense.weight.data()._t
        wq = ops.weight_fake_quant(w.contiguous(), 1, 8)
        return (X._t @ wq.t() + dense.bias.data()._t).cpu().numpy()
    calls = []
    real = ops.weight_fake_quant

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)
    ops.weight_fake_quant = spy
    try:
        y0 = net(X).asnumpy()
        y0b = net(X).asnumpy()
        assert len(calls) == 1                                       # second forward: cached
        dense.weight.data()._t.mul_(1.7)                             # in-place update (optimiser step)
        y1 = net(X).asnumpy()
        assert len(calls) == 2
        dense.weight.set_data(mx.nd.array(rng.standard_normal((8, 16)).astype(np.float32), ctx=gpu))
        y2 = net(X).asnumpy()
        assert len(calls) == 3
    finally:
        ops.weight_fake_quant = real
    np.testing.assert_array_equal(y0, y0b)
    np.testing.assert_allclose(y2, expect(), rtol=1e-5, atol=1e-5)
    assert not np.allclose(y0, y1)


@pytest.mark.parametrize("legacy", [False, True], ids=["one-call", "legacy-gemm"])
def test_quantized_conv2d_is_exact_beyond_the_fp32_accumulator_limit(gpu, legacy, monkeypatch):
    """nn.Conv2D(quantized=True): with 128 input channels, 3x3 filters and saturated codes the integer
    accumulator reaches 1152 * 255 * 127 = 3.7e7 > 2^24, where an fp32 accumulation (the reference's own `dot`,
    nn/quantized_conv.py:140-144) is no longer exact.  The block must return in_scale * w_scale times the EXACT integer
    correlation - through the one-call entry point (fq_qconv2d_forward; this geometry: the exact direct kernel) and through
    the round-3 formulation kept behind FQ_QCONV_LEGACY=1 (im2col of the codes + the int8 matrix-core GEMM)."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.nn import Conv2D
    from oracle import host as H
    rng = np.random.default_rng(9)
    x = np.full((2, 128, 6, 6), 3.0, np.float32)
    x[:, :, ::2, ::3] = rng.uniform(2.5, 3.0, x[:, :, ::2, ::3].shape).astype(np.float32)
    x[0, 0, 0, 0] = 0.0                                              # uint8 range [0, 3]
    w = np.full((40, 128, 3, 3), 0.5, np.float32)
    w[::3] = rng.uniform(0.45, 0.5, w[::3].shape).astype(np.float32)
    w[1, 0, 0, 0] = -0.5                                             # int8 symmetric range 0.5
    conv = Conv2D(40, 3, 1, 0, in_channels=128, use_bias=False, quantized=True, input_dtype='uint8',
                  weight_dtype='int8')
    conv.initialize(ctx=gpu)
    conv.weight.set_data(mx.nd.array(w, ctx=gpu))
    from quantization.mxnet_amd import ops
    monkeypatch.setenv("FQ_QCONV_LEGACY", "1" if legacy else "0")
    seen, fused = [], []
    real, real_q = ops.gemm_i8_codes, ops.qconv2d

    def spy(*a, **k):
        out = real(*a, **k)
        seen.append(out.detach().cpu().numpy().astype(np.int64))
        return out

    def spy_q(*a, **k):
        fused.append(1)
        return real_q(*a, **k)
    ops.gemm_i8_codes, ops.qconv2d = spy, spy_q
    try:
        y = conv(mx.nd.array(x, ctx=gpu)).asnumpy()
    finally:
        ops.gemm_i8_codes, ops.qconv2d = real, real_q
    in_scale = np.float32(np.float32(3.0 - 0.0) / np.float32(255))
    w_scale = np.float32(np.float32(0.5) / np.float32(127))
    xc = np.round(x / in_scale).astype(np.int64)
    wc = np.round(w / w_scale).astype(np.int64)
    assert xc.max() == 255 and wc.max() == 127
    want = np.zeros((2, 40, 4, 4), np.int64)
    for i in range(4):
        for j in range(4):
            want[:, :, i, j] = np.einsum("ncij,ocij->no", xc[:, :, i:i + 3, j:j + 3], wc)
    assert np.abs(want).max() > 2 ** 24
    if legacy:
        assert len(seen) == 1 and not fused                           # the block went through the int8 matrix-core GEMM
        np.testing.assert_array_equal(seen[0].reshape(want.shape), want)  # exact integers
    else:
        assert len(fused) == 1 and not seen
    # fp32(exact integer) * fp32(in_scale * w_scale): what the oracle's block returns, bit for bit
    np.testing.assert_array_equal(y, H.qconv2d_forward(x, w, None, (1, 1), (0, 0), 1))
    np.testing.assert_array_equal(y, (want.astype(np.float32) * np.float32(in_scale * w_scale)).astype(np.float32))


@pytest.mark.parametrize("model,offline", [("mobilenet1.0", False), ("resnet50_v1", True)], ids=["mobilenet1.0-online", "resnet50_v1-offline"])
def test_eval_head_counts_in_the_classifier_launch(gpu, model, offline):
    """quantize/fuse.py EvalHead: with the labels bound, the classifier's launch (fq_dense_i8_eval) adds exactly what
    `ops.eval_counters` adds to the same logits; without labels the forward counts nothing; logits do not change."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.quantize import fuse
    net = _build(model, 1000, gpu)
    rng = np.random.default_rng(3)
    batch, hw = 6, 64
    if offline:
        net(mx.nd.array(rng.standard_normal((batch, 3, hw, hw)).astype(np.float32), ctx=gpu))
        net.update_ema()
    net.fix_params()
    net.quantize_input(enable=True, online=not offline)
    assert fuse.fuse_inference(net) > 0
    dev = gpu.torch_device
    c_head = torch.zeros(2002, device=dev)
    c_ref = torch.zeros(2002, device=dev)
    head = fuse.eval_head(net, c_head)
    assert head is not None and getattr(head.block, "_fq_dense_int8", False)
    for i in range(3):
        X = mx.nd.array(rng.standard_normal((batch, 3, hw, hw)).astype(np.float32), ctx=gpu)
        labels = torch.from_numpy(rng.integers(0, 1000, batch).astype(np.int64)).to(dev)
        ref = net(X)._t.clone()
        assert not head.take()                                       # no labels bound: nothing counted
        if i == 1:
            labels[0] = int(ref[0].argmax())                          # at least one correct prediction
        ops.eval_counters(ref, labels, c_ref)
        head.labels = labels
        out = net(X)._t
        assert head.take() and head.labels is None
        assert torch.equal(out, ref)
    assert torch.equal(c_head, c_ref) and float(c_ref[1]) == 3 * batch and float(c_ref[0]) >= 1
    head.release()
    assert "_fq_eval_head" not in head.block.__dict__
    fuse.unfuse(net)


@pytest.mark.parametrize("groups,cin,expect_gemm", [(2, 256, True), (4, 64, False), (32, 32, False)],
                         ids=["2-groups-long-dot", "4-groups-short-dot", "depthwise"])
@pytest.mark.parametrize("legacy", [False, True], ids=["one-call", "legacy"])
def test_quantized_grouped_conv2d_is_exact(gpu, groups, cin, expect_gemm, legacy, monkeypatch):
    """nn.Conv2D(quantized=True, groups > 1) with int8 weights (reference: a Python loop over the groups, nn/quantized_conv.py:
    129-151): saturated codes, dot lengths below and beyond the point where an fp32 accumulator stops being exact (128 * 9 =
    1152 terms pass 2^24) - the block returns in_scale * w_scale times the EXACT integer correlation: through the one-call
    entry point (grouped 3x3 without padding: the exact direct kernel) and through the round-3 formulation
    (FQ_QCONV_LEGACY=1: the int8 matrix-core GEMM group by group for long dots, one fp32-exact grouped product otherwise)."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.nn import Conv2D
    from oracle import host as H
    rng = np.random.default_rng(groups + cin)
    cout = 2 * groups
    x = np.full((2, cin, 5, 5), 3.0, np.float32)
    x[:, :, ::2, ::3] = rng.uniform(2.5, 3.0, x[:, :, ::2, ::3].shape).astype(np.float32)
    x[0, 0, 0, 0] = 0.0                                              # uint8 range [0, 3]
    w = np.full((cout, cin // groups, 3, 3), 0.5, np.float32)
    w[::2] = rng.uniform(0.45, 0.5, w[::2].shape).astype(np.float32)
    w[1, 0, 0, 0] = -0.5                                             # int8 symmetric range 0.5
    conv = Conv2D(cout, 3, 1, 0, in_channels=cin, groups=groups, use_bias=False, quantized=True, input_dtype='uint8',
                  weight_dtype='int8')
    conv.initialize(ctx=gpu)
    conv.weight.set_data(mx.nd.array(w, ctx=gpu))
    monkeypatch.setenv("FQ_QCONV_LEGACY", "1" if legacy else "0")
    calls = []
    real = ops.gemm_i8_codes

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)
    ops.gemm_i8_codes = spy
    try:
        y = conv(mx.nd.array(x, ctx=gpu)).asnumpy()
    finally:
        ops.gemm_i8_codes = real
    if legacy:
        assert (len(calls) == groups) if expect_gemm else (len(calls) == 0)
    else:
        assert not calls
    in_scale = np.float32(np.float32(3.0 - 0.0) / np.float32(255))
    w_scale = np.float32(np.float32(0.5) / np.float32(127))
    xc = np.round(x / in_scale).astype(np.int64)
    wc = np.round(w / w_scale).astype(np.int64)
    cg, og = cin // groups, cout // groups
    want = np.zeros((2, cout, 3, 3), np.int64)
    for g in range(groups):
        for i in range(3):
            for j in range(3):
                want[:, g * og:(g + 1) * og, i, j] = np.einsum("ncij,ocij->no", xc[:, g * cg:(g + 1) * cg, i:i + 3, j:j + 3],
                                                               wc[g * og:(g + 1) * og])
    assert (np.abs(want).max() > 2 ** 24) == expect_gemm
    np.testing.assert_array_equal(y, H.qconv2d_forward(x, w, None, (1, 1), (0, 0), groups))
    np.testing.assert_array_equal(y, (want.astype(np.float32) * np.float32(in_scale * w_scale)).astype(np.float32))


@pytest.mark.parametrize("model,offline", [("mobilenet1.0", False), ("resnet50_v1", True)], ids=["mobilenet1.0-online", "resnet50_v1-offline"])
def test_forwards_in_flight_on_several_streams_equal_sequential_forwards(gpu, model, offline):
    """An evaluation loop may keep several independent batches in flight, one HIP stream each (bench.py --streams): a forward
    on a side stream takes its statistic arena, its batch-statistic slots and its workspaces per stream, so the forwards
    overlap on the device and give what they give one after the other - logits bit for bit, counters equal - and leave the
    blocks' `current_input_max` (the calibration state of the default stream) alone."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.quantize import fuse
    net = _build(model, 1000, gpu)
    rng = np.random.default_rng(11)
    batch, hw = 8, 64
    xs = [mx.nd.array(rng.standard_normal((batch, 3, hw, hw)).astype(np.float32) * (1 + i), ctx=gpu) for i in range(6)]
    if offline:
        net(xs[0])
        net.update_ema()
    net.fix_params()
    net.quantize_input(enable=True, online=not offline)
    assert fuse.fuse_inference(net) > 0
    dev = gpu.torch_device
    labels = [torch.from_numpy(rng.integers(0, 1000, batch).astype(np.int64)).to(dev) for _ in xs]
    c_seq = torch.zeros(2002, device=dev)
    c_par = torch.zeros(2002, device=dev)
    head = fuse.eval_head(net, c_seq)
    ref = []
    for x, lb in zip(xs, labels):                       # one after the other on the default stream
        head.labels = lb
        ref.append(net(x)._t.clone())
        assert head.take()
    blocks = net.collect_quantized_blocks()
    cur_before = [float(b.current_input_max) for b in blocks if hasattr(b, "current_input_max")]
    torch.cuda.synchronize()
    head.counters = c_par
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    outs = []
    for rep in range(3):                                # ... and with three in flight, three times over
        outs = []
        for i, (x, lb) in enumerate(zip(xs, labels)):
            with torch.cuda.stream(streams[i % 3]), ops.batches_in_flight():
                head.labels = lb
                outs.append(net(x)._t)
                assert head.take()
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            assert torch.equal(o, r)
    assert torch.equal(c_par, 3 * c_seq)
    cur_after = [float(b.current_input_max) for b in blocks if hasattr(b, "current_input_max")]
    assert cur_after == cur_before
    head.release()
    fuse.unfuse(net)


@pytest.mark.parametrize("fused", [False, True], ids=["plain", "fused"])
def test_calibration_on_a_non_default_stream_is_an_ordinary_calibration(gpu, fused):
    """whether a forward counts for calibration is NOT inferred from the HIP stream.  `net(X); net.update_ema()`
    under `torch.cuda.stream(s)` (no `ops.batches_in_flight()` declaration) gives the thresholds of the same loop on the
    default stream, bit for bit; inside the declaration `update_ema` refuses."""
    from quantization.mxnet_amd import mx, ops
    from quantization.mxnet_amd.quantize import fuse
    rng = np.random.default_rng(5)
    xs = [mx.nd.array(rng.standard_normal((4, 3, 64, 64)).astype(np.float32) * (1 + i), ctx=gpu) for i in range(3)]

    def thresholds(stream):
        net = _build("mobilenet1.0", 10, gpu)
        if fused:
            fuse.fuse_inference(net)
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            for x in xs:
                net(x)
                net.update_ema()
        torch.cuda.synchronize()
        return [float(b.input_max.data()._t.item()) for b in net.collect_quantized_blocks() if hasattr(b, "input_max")]

    torch.cuda.synchronize()
    want = thresholds(None)
    side = torch.cuda.Stream(gpu.torch_device)
    side.wait_stream(torch.cuda.current_stream(gpu.torch_device))
    got = thresholds(side)
    assert all(w > 0 for w in want)
    assert got == want
    net = _build("mobilenet1.0", 10, gpu)
    with ops.batches_in_flight():
        with pytest.raises(RuntimeError, match="batches_in_flight"):
            net.update_ema()
    assert not ops.in_flight()


def test_dense_on_an_unflattened_input_against_the_reference_fixture(gpu, golden):
    """G12 (tests/golden/g12_dense_unflattened.npz: the reference's `_dense_forward` on (N, C, H, W) inputs): the converted Dense on
    the GPU quantises with the statistic of convert_dense.py:41 - max over axis 1 only, mean over N * H * W values - online and
    with a stored threshold."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.quantize import convert
    from quantization.mxnet_amd.quantize.initialize import qparams_init
    g = golden("g12_dense_unflattened")
    for tag in sorted({k.split("/")[0] for k in g}):
        x = g[tag + "/x"]
        net = nn.HybridSequential()
        net.add(nn.Dense(5, in_units=int(np.prod(x.shape[1:]))))
        net.initialize(mx.init.Xavier())
        convert.convert_model(net, convert_fn={nn.Dense: convert.gen_dense_converter(quantize_input=True, input_signed=tag.endswith("_s"))})
        qparams_init(net)
        net.collect_params().reset_ctx(gpu)
        blk = net[0]
        seen = []
        real = blk.origin_forward
        blk.origin_forward = lambda F, xq, w, b=None, _r=real: (seen.append(xq.asnumpy()), _r(F, xq, w, b))[1]
        X = mx.nd.array(x, ctx=gpu)
        net(X)
        assert np.float32(float(blk.current_input_max)) == g[tag + "/online_max"], tag
        assert np.array_equal(seen[-1], g[tag + "/online_y"]), tag
        blk.input_max.set_data(mx.nd.array(np.asarray([g[tag + "/offline_thr"]], np.float32), ctx=gpu))
        net.quantize_input(enable=True, online=False)
        net(X)
        assert np.array_equal(seen[-1], g[tag + "/offline_y"]), tag + " offline"
        assert np.float32(float(blk.current_input_max)) == g[tag + "/offline_curmax"], tag
