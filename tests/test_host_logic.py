"""Host logic of the product (converters, fixed_params state machine, EMA arena, KL collection loop, nn.Conv2D
plumbing) on CPU, with the HIP entry points replaced by the oracle (oracle/patch.py) — and compared with the goldens
produced by the reference's own convert.py / convert_conv2d.py / initialize.py running over the same tiny net."""
import numpy as np
import pytest
import torch

from oracle.patch import oracle_ops
from quantization.mxnet_amd import mx
from quantization.mxnet_amd.mx.gluon import nn
from quantization.mxnet_amd.mx.gluon.block import reset_naming
from quantization.mxnet_amd.quantize import convert
from quantization.mxnet_amd.quantize.initialize import qparams_init


def tiny_net(params):
    reset_naming()
    net = nn.HybridSequential(prefix="tiny_")
    with net.name_scope():
        net.add(nn.Conv2D(8, 3, padding=1, in_channels=3, use_bias=False), nn.Activation("relu"),
                nn.Conv2D(8, 3, padding=1, groups=8, in_channels=8, use_bias=True), nn.Activation("relu"),
                nn.Conv2D(12, 1, in_channels=8, use_bias=False), nn.Activation("relu"),
                nn.GlobalAvgPool2D(), nn.Flatten(), nn.Dense(5, in_units=12))
    net.initialize()
    for name, p in net.collect_params().items():
        p.set_data(mx.nd.array(params[name]))
    return net


def build(g, tag, quant_type, wt):
    params = {k.split("/param/")[1]: v for k, v in g.items() if k.startswith(tag + "/param/")}
    net = tiny_net(params)
    convert_fn = {nn.Conv2D: convert.gen_conv2d_converter(quant_type=quant_type, weight_width=wt),
                  nn.Dense: convert.gen_dense_converter(quant_type=quant_type, weight_width=wt),
                  nn.Activation: None, nn.BatchNorm: None}
    convert.convert_model(net, exclude=[net[0]], convert_fn=convert_fn)
    qparams_init(net)
    return net


def close(a, b, what):
    # conv/FC themselves run in torch on both sides; allow fp32 reassociation noise there, nothing more
    np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6, err_msg=what)


@pytest.mark.parametrize("tag,quant_type,wt", [("layer_w8", "layer", 8), ("channel_w4", "channel", 4)])
def test_cli_state_machine_matches_reference(golden, tag, quant_type, wt):
    g = golden("g7_g9_ema_state")
    with oracle_ops():
        net = build(g, tag, quant_type, wt)
        blocks = net.collect_quantized_blocks()
        assert len(blocks) == int(g[tag + "/n_blocks"]) == 3
        assert net[0] not in blocks and not hasattr(net[0], "quantize_args")          # excluded by identity
        assert [getattr(b, "fixed_params", -9) for b in blocks] == [-1, -1, -9]       # Dense has no fixed_params
        xs = g[tag + "/xs"]
        # A. naive calibration: online + update_ema
        net.quantize_input(enable=True, online=True)
        for i, x in enumerate(xs):
            y = net(mx.nd.array(x))
            net.update_ema()
            cur = np.asarray([float(b.current_input_max) for b in blocks], np.float32)
            ema = np.asarray([b.input_max.data().asscalar() for b in blocks], np.float32)
            np.testing.assert_array_equal(cur, g[tag + "/calib_cur"][i], "current_input_max step %d" % i)
            np.testing.assert_array_equal(ema, g[tag + "/calib_ema"][i], "input_max EMA step %d" % i)
            close(y.asnumpy(), g[tag + "/calib_logits"][i], "calibration logits %d" % i)
        # the EMA state is one contiguous vector and the Parameters are views into it
        arena = net.calibration_arena()
        assert arena.state.numel() == 3
        for i, b in enumerate(blocks):
            assert b.input_max.data()._t.data_ptr() == arena.state.data_ptr() + 4 * i
        # B. freeze + offline
        net.fix_params()
        net.quantize_input(enable=True, online=False)
        np.testing.assert_array_equal([getattr(b, "fixed_params", -9) for b in blocks], g[tag + "/fixed_before"])
        close(net(mx.nd.array(xs[0])).asnumpy(), g[tag + "/offline_logits0"], "offline logits 0")
        np.testing.assert_array_equal([getattr(b, "fixed_params", -9) for b in blocks], g[tag + "/fixed_after"])
        close(net(mx.nd.array(xs[1])).asnumpy(), g[tag + "/offline_logits1"], "offline logits 1")
        for name, p in net.collect_params().items():
            np.testing.assert_array_equal(p.data().asnumpy(), g["%s/frozen/%s" % (tag, name)],
                                          "frozen parameter " + name)
        # C. disable_quantize / D. online on frozen net / E. input quantisation off
        net.disable_quantize()
        close(net(mx.nd.array(xs[2])).asnumpy(), g[tag + "/disabled_logits"], "disabled")
        net.enable_quantize()
        net.quantize_input(enable=True, online=True)
        close(net(mx.nd.array(xs[3])).asnumpy(), g[tag + "/online_frozen_logits"], "online on frozen")
        net.quantize_input(enable=False)
        close(net(mx.nd.array(xs[4])).asnumpy(), g[tag + "/noinput_logits"], "no input quant")


def test_converter_api_surface():
    conv = nn.Conv2D(4, 3, in_channels=2)
    conv.initialize()
    convert.gen_conv2d_converter(weight_width=4, quant_type="group", input_signed=True, input_width=6,
                                 wino_quantize="F43")(conv)
    qa = conv.quantize_args
    assert (qa.wt_width, qa.quant_type, qa.in_signed, qa.in_width, qa.wino_quantize, qa.fake_bn,
            qa.quantize_input) == (4, "group", True, 6, "F43", False, True)
    assert conv.fixed_params == -1 and conv.enable_quantize and conv.quantize_input
    assert conv.quantize_input_offline is False and conv.current_input_max == 0.
    assert conv.input_max.shape == (1,) and "input_max" in conv._reg_params
    assert callable(conv.origin_forward)
    with pytest.raises(AssertionError):
        convert.gen_conv2d_converter(wino_quantize="F99")
    with pytest.raises(AssertionError):
        convert.gen_dense_converter()(conv)
    dense = nn.Dense(3, in_units=4)
    convert.gen_dense_converter(quant_type="group")(dense)
    assert dense.quantize_args.quant_type == "channel" and not hasattr(dense, "fixed_params")
    act = nn.Activation("relu")
    convert.gen_act_converter(width=4)(act)
    assert act.quantize_args.width == 4 and act.act_max.shape == (1,)
    assert set(convert.default_convert_fn) == {nn.Conv2D, nn.Dense, nn.Activation, nn.BatchNorm}


def test_dispatch_is_by_exact_type_with_custom_override_and_exclude():
    class MyConv(nn.Conv2D):
        pass
    reset_naming()
    net = nn.HybridSequential()
    a, b, c = nn.Conv2D(2, 1, in_channels=2), MyConv(2, 1, in_channels=2), nn.Conv2D(2, 1, in_channels=2)
    net.add(a, b, c)
    marker = []
    convert.convert_model(net, exclude=[c], custom_fn={b: lambda m: marker.append(m)})
    assert hasattr(a, "quantize_args") and not hasattr(b, "quantize_args") and not hasattr(c, "quantize_args")
    assert marker == [b]
    assert net.collect_quantized_blocks() == [a]
    for meth in ("update_ema", "collect_quantized_blocks", "quantize_input", "enable_quantize", "disable_quantize",
                 "fix_params"):
        assert callable(getattr(net, meth))


def test_group_quant_on_general_grouped_conv_raises_like_the_reference_broadcast():
    conv = nn.Conv2D(8, 1, groups=2, in_channels=4, use_bias=False)
    conv.initialize()
    convert.gen_conv2d_converter(quant_type="group", quantize_input=False)(conv)
    with oracle_ops(), pytest.raises(ValueError, match="broadcast"):
        conv(mx.nd.array(np.ones((1, 4, 2, 2), np.float32)))


def test_product_on_cpu_without_oracle_raises():
    from quantization.mxnet_amd._lib import FakeQuantError
    conv = nn.Conv2D(4, 1, in_channels=2)
    conv.initialize()
    convert.gen_conv2d_converter()(conv)
    conv.input_max.initialize(mx.initializer.Constant(0))
    with pytest.raises(FakeQuantError, match="no CPU fallback"):
        conv(mx.nd.array(np.ones((1, 2, 3, 3), np.float32)))


def test_collect_feature_maps_and_kl_loop(golden):
    """The calibration driver of simulate_quantization.py:294-315 against the reference's collect_feature_maps."""
    from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps, kl_calibrate
    g = golden("g3_collect")

    class Blk(nn.HybridBlock):
        def __init__(self, k):
            super().__init__()
            self.k = k
            self.quantize_args = True

        def forward(self, x):
            return x

    class Net(nn.HybridBlock):
        def __init__(self):
            super().__init__()
            self.b = [Blk(0), Blk(1), Blk(2)]

        def collect_quantized_blocks(self):
            return self.b

        def forward(self, X):
            for k, b in enumerate(self.b):
                fm = mx.nd.relu(X) * float(k + 1)
                if k == 2:
                    fm = mx.nd.NDArray(fm._t[:, :, ::2, ::2].contiguous())
                b(fm)
            return X
    net = Net()
    loader = [(mx.nd.array(b), None) for b in g["batches"]]
    with oracle_ops():
        hists, maxes = collect_feature_maps(net, 2048, loader, mx.cpu())
        for k, b in enumerate(net.b):
            np.testing.assert_array_equal(hists[b], g["hist%d" % k])
            assert maxes[b] == g["fm_max%d" % k] and hists[b].dtype == np.float32
        assert all(len(b._forward_hooks) == 0 for b in net.b)            # hooks detached (:111-112)
        gk = golden("g2_kl")
        assert kl_calibrate(gk["sparse/hist_b256"], 16, 16, 256) == int(gk["sparse/best_b256_L16"])
        with pytest.raises(AssertionError, match="min_bins should be greater than levels"):
            kl_calibrate(gk["sparse/hist_b256"], 32, 16, 256)

        class RawNet(Net):
            def forward(self, X):
                self.b[0](X)
                return X
        raw = RawNet()
        raw.b = raw.b[:1]
        bad = [(mx.nd.array(np.float32([[[[1.0, -0.5], [2.0, 0.0]]]])), None)]
        with pytest.raises(AssertionError, match="Activation should >=0"):
            collect_feature_maps(raw, 16, bad, mx.cpu())
        with pytest.raises(AssertionError, match="all zero-value"):
            collect_feature_maps(raw, 16, [(mx.nd.array(np.zeros((1, 1, 2, 2), np.float32)), None)], mx.cpu())


@pytest.mark.parametrize("use_bias", [0, 1])
@pytest.mark.parametrize("groups", [1, 2])
def test_nn_conv2d_three_way(golden, use_bias, groups):
    """reference tests/test_quantized_conv.py:36-57, asserted: int-code conv == the reference's result."""
    from quantization.mxnet_amd import nn as qnn
    g = golden("g8_quantized_conv")
    tag = "conv_b%d_g%d" % (use_bias, groups)
    with oracle_ops():
        for quantized in (False, True):
            c = qnn.Conv2D(10, 3, 1, 1, in_channels=2, groups=groups, use_bias=bool(use_bias), quantized=quantized,
                           input_dtype="uint8", weight_dtype="int8")
            c.initialize()
            c.weight.set_data(mx.nd.array(g[tag + "/w"]))
            if use_bias:
                c.bias.set_data(mx.nd.array(g[tag + "/b"]))
            y = c(mx.nd.array(g[tag + "/x"])).asnumpy()
            if quantized:
                np.testing.assert_array_equal(y, g[tag + "/y_int"])
            else:
                np.testing.assert_allclose(y, g[tag + "/y_float"], rtol=1e-5, atol=1e-5)


def test_ste_function_api(golden):
    g = golden("g4_activation")
    tag = "conv_4x8x7x7_u_w8"
    with oracle_ops():
        ste = convert.LinearQuantizeSTE(g[tag + "/online_scale"], g[tag + "/online_max"], 0.0)
        y = ste(mx.nd.array(g[tag + "/x"]))
        np.testing.assert_array_equal(y.asnumpy(), g[tag + "/online_y"])
        dy = mx.nd.array(np.ones(3, np.float32))
        assert ste.backward(dy) is dy
        assert convert.LinearQuantizeSTE(1.0).clip_min == 0.0


def test_model_zoo_layouts_match_cli_exclusions():
    from quantization.mxnet_amd.mx.gluon.model_zoo import get_model
    np.random.seed(7)
    net = get_model("mobilenet1.0", classes=1000)
    convert.convert_model(net, exclude=[net.features[0], net.features[1]],
                          convert_fn={nn.Conv2D: convert.gen_conv2d_converter(), nn.Dense: convert.gen_dense_converter(),
                                      nn.Activation: None, nn.BatchNorm: None})
    assert len(net.collect_quantized_blocks()) == 27
    net = get_model("resnet50_v1", classes=1000)
    convert.convert_model(net, exclude=[net.features[0], net.features[1]])
    assert len(net.collect_quantized_blocks()) == 53
    net = get_model("cifar_resnet20_v1", classes=10)
    convert.convert_model(net, exclude=[net.features[0], net.features[1], net.features[2][0].body[0],
                                        net.features[2][0].body[1]])
    assert len(net.collect_quantized_blocks()) == 20
    net = get_model("mobilenetv2_1.0", classes=1000)
    convert.convert_model(net, exclude=[net.features[0], net.features[1], net.output[0]])
    assert len(net.collect_quantized_blocks()) == 52


# ---- G10: fake-BN (--merge-bn) and one-shot merge_bn against the reference's own code ---------------------------------
def _bn_net(params):
    reset_naming()
    net = nn.HybridSequential(prefix="bnnet_")
    with net.name_scope():
        net.add(nn.Conv2D(8, 3, padding=1, in_channels=3, use_bias=False), nn.BatchNorm(in_channels=8), nn.Activation("relu"),
                nn.Conv2D(8, 3, padding=1, groups=8, in_channels=8, use_bias=False), nn.BatchNorm(in_channels=8),
                nn.Activation("relu"),
                nn.Conv2D(12, 1, in_channels=8, use_bias=True), nn.BatchNorm(in_channels=12), nn.Activation("relu"),
                nn.GlobalAvgPool2D(), nn.Flatten(), nn.Dense(5, in_units=12))
    net.initialize()
    for name, p in net.collect_params().items():
        p.set_data(mx.nd.array(params[name]))
    return net


def _g10_params(g):
    return {k.split("param/")[1]: v for k, v in g.items() if k.startswith("param/")}


def run_fake_bn_flow(g, ctx=None):
    net = _bn_net(_g10_params(g))
    convert_fn = {nn.Conv2D: convert.gen_conv2d_converter(fake_bn=True, input_signed=True),
                  nn.Dense: convert.gen_dense_converter(input_signed=True),
                  nn.Activation: None, nn.BatchNorm: convert.bypass_bn}
    convert.convert_model(net, exclude=[net[0], net[1]], convert_fn=convert_fn)
    qparams_init(net)
    if ctx is not None:
        net.collect_params().reset_ctx(ctx)
    xs = g["xs"]
    arr = (lambda a: mx.nd.array(a, ctx=ctx)) if ctx is not None else mx.nd.array
    net.quantize_input(enable=True, online=True)
    out = {"calib": net(arr(xs[0])).asnumpy()}
    net.fix_params()
    out["frozen0"] = net(arr(xs[1])).asnumpy()
    out["frozen1"] = net(arr(xs[2])).asnumpy()
    out["params"] = {name: p.data().asnumpy() for name, p in net.collect_params().items()}
    return out


def test_fake_bn_merge_bn_flag_matches_reference(golden):
    g = golden("g10_fake_bn")
    with oracle_ops():
        out = run_fake_bn_flow(g)
    close(out["calib"], g["fakebn/calib_logits"], "fake-bn calibration logits")
    close(out["frozen0"], g["fakebn/frozen_logits0"], "fake-bn frozen logits 0")
    close(out["frozen1"], g["fakebn/frozen_logits1"], "fake-bn frozen logits 1")
    for name, v in out["params"].items():
        np.testing.assert_array_equal(v, g["fakebn/frozen/" + name], "frozen " + name)
    assert "bnnet_conv1_bias" in out["params"]          # bias created by qparams_init (initialize.py:63-70)


def test_merge_bn_matches_reference(golden, capsys):
    from quantization.mxnet_amd.quantize.freeze import merge_bn
    g = golden("g10_fake_bn")
    net = _bn_net(_g10_params(g))
    x = mx.nd.array(g["xs"][0])
    before = net(x).asnumpy()
    merge_bn(net)
    assert "Merge bnnet_batchnorm0 to bnnet_conv0" in capsys.readouterr().out
    after = net(x).asnumpy()
    close(before, g["merge/logits_before"], "before")
    close(after, g["merge/logits_after"], "after")
    for name, p in net.collect_params().items():
        np.testing.assert_allclose(p.data().asnumpy(), g["merge/param/" + name], rtol=1e-6, atol=1e-7, err_msg=name)
    conv = net[0]
    convert.gen_conv2d_converter(fake_bn=True)(conv)
    with pytest.raises(AssertionError, match="fake bn"):
        merge_bn(net)


def test_scale_table_export(tmp_path):
    """quantize/freeze/scale_table.py: per-channel weight scales of the BN-FOLDED weights and one
    input scale per layer, ncnn table layout (README.md:267-274 of the reference fixes the content)."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.mx.gluon.block import reset_naming
    from quantization.mxnet_amd.quantize import convert
    from quantization.mxnet_amd.quantize.initialize import qparams_init
    from quantization.mxnet_amd.quantize.freeze import export_scale_table
    reset_naming()
    rng = np.random.default_rng(4)
    net = nn.HybridSequential()
    net.add(nn.Conv2D(6, 3, 1, 1, use_bias=False, in_channels=3), nn.BatchNorm(in_channels=6), nn.Activation("relu"),
            nn.GlobalAvgPool2D(), nn.Flatten(), nn.Dense(4, in_units=6))
    net.initialize()
    conv, bn, dense = net[0], net[1], net[5]
    w = rng.standard_normal((6, 3, 3, 3)).astype(np.float32)
    g = rng.uniform(0.5, 2.0, 6).astype(np.float32)
    var = rng.uniform(0.5, 2.0, 6).astype(np.float32)
    conv.weight.set_data(mx.nd.array(w))
    bn.gamma.set_data(mx.nd.array(g))
    bn.running_var.set_data(mx.nd.array(var))
    wd = rng.standard_normal((4, 6)).astype(np.float32)
    dense.weight.set_data(mx.nd.array(wd))
    convert.convert_model(net, convert_fn={nn.Conv2D: convert.gen_conv2d_converter(quant_type="channel", fake_bn=True),
                                           nn.Dense: convert.gen_dense_converter(quant_type="channel"),
                                           nn.BatchNorm: convert.bypass_bn})
    qparams_init(net)
    conv.input_max.set_data(mx.nd.array(np.float32([2.5])))
    dense.input_max.set_data(mx.nd.array(np.float32([0.8])))
    path = str(tmp_path / "table.txt")
    entries = export_scale_table(net, path, weight_width=8, input_width=8, json_path=path + ".json")
    folded = w * (g / np.sqrt(var + np.float32(1e-10))).reshape(-1, 1, 1, 1)
    by = {e["name"]: e for e in entries}
    np.testing.assert_allclose(by[conv.name + "_param_0"]["scales"], 127.0 / np.abs(folded).reshape(6, -1).max(axis=1),
                               rtol=1e-6)
    np.testing.assert_allclose(by[dense.name + "_param_0"]["scales"], 127.0 / np.abs(wd).max(axis=1), rtol=1e-6)
    np.testing.assert_allclose(by[conv.name]["scales"], [127.0 / 2.5], rtol=1e-6)
    np.testing.assert_allclose(by[dense.name]["scales"], [127.0 / 0.8], rtol=1e-6)
    lines = open(path).read().strip().split("\n")
    assert [l.split()[0] for l in lines] == [conv.name + "_param_0", dense.name + "_param_0", conv.name, dense.name]
    assert len(lines[0].split()) == 1 + 6 and len(lines[2].split()) == 2
    import json
    assert json.load(open(path + ".json"))[0]["kind"] == "weight"
    # the block's own widths when none are forced: unsigned 8-bit input -> 255 levels
    own = {e["name"]: e for e in export_scale_table(net)}
    np.testing.assert_allclose(own[conv.name]["scales"], [255.0 / 2.5], rtol=1e-6)


def test_weight_fake_quant_is_reused_until_the_parameter_changes():
    """The reference re-quantises every weight on every forward (convert_conv2d.py:68-99, convert_dense.py:52-63); the product
    keeps the result while the parameter is the same tensor in the same in-place version: same values, one computation.  An
    in-place update (an optimiser step), `set_data` and a recording forward each compute afresh."""
    from quantization.mxnet_amd import ops
    from quantization.mxnet_amd.mx import autograd
    rng = np.random.default_rng(5)
    params = {"tiny_conv0_weight": rng.standard_normal((8, 3, 3, 3)).astype(np.float32),
              "tiny_conv1_weight": rng.standard_normal((8, 1, 3, 3)).astype(np.float32),
              "tiny_conv1_bias": rng.standard_normal(8).astype(np.float32),
              "tiny_conv2_weight": rng.standard_normal((12, 8, 1, 1)).astype(np.float32),
              "tiny_dense0_weight": rng.standard_normal((5, 12)).astype(np.float32),
              "tiny_dense0_bias": rng.standard_normal(5).astype(np.float32)}
    x = mx.nd.array(np.abs(rng.standard_normal((2, 3, 6, 6))).astype(np.float32))
    with oracle_ops():
        net = tiny_net(params)
        convert_fn = {nn.Conv2D: convert.gen_conv2d_converter(), nn.Dense: convert.gen_dense_converter(),
                      nn.Activation: None, nn.BatchNorm: None}
        convert.convert_model(net, exclude=[net[0]], convert_fn=convert_fn)
        qparams_init(net)
        calls = []
        real = ops.weight_fake_quant

        def counting(w, *a, **k):
            calls.append(tuple(w.shape))
            return real(w, *a, **k)
        ops.weight_fake_quant = counting
        try:
            y0 = net(x).asnumpy()
            assert len(calls) == 3                                    # two converted convolutions + the Dense
            y1 = net(x).asnumpy()
            assert len(calls) == 3 and np.array_equal(y0, y1)         # nothing re-quantised
            net[4].weight.data()._t.mul_(2.0)                         # in-place update of ONE parameter
            y2 = net(x).asnumpy()
            assert len(calls) == 4 and calls[-1] == (12, 8, 1, 1) and not np.array_equal(y2, y0)
            net[2].weight.set_data(mx.nd.array(params["tiny_conv1_weight"] * 0.5))
            net(x)
            assert len(calls) == 5 and calls[-1] == (8, 1, 3, 3)
            with autograd.record():                                   # a recording forward owns its straight-through links
                net(x)
            assert len(calls) == 7                                    # (the Dense keeps its cache: its link is made per forward)
            net(x)
            assert len(calls) == 7
        finally:
            ops.weight_fake_quant = real


def test_quantized_mobilenet_and_its_producer_fusion_on_the_oracle():
    """nn/quantized_mobilenet.py (the reference's tests/models/quantized_mobilenet.py) + nn/fuse.py with the oracle standing in
    for the library: 26 quantised convolutions; fused, every one takes BatchNorm + ReLU into its store and its range from the
    producer's statistic (the oracle checks that statistic against the tensor); `unfuse` restores the blocks."""
    from quantization.mxnet_amd import ops
    from quantization.mxnet_amd.nn import quantized_mobilenet as QM, fuse as qfuse
    np.random.seed(5)
    net = QM.MobileNet(1.0, classes=10)
    net.initialize(mx.init.Xavier(magnitude=2.0))
    x = mx.nd.array(np.random.default_rng(1).standard_normal((2, 3, 32, 32)).astype(np.float32))
    with oracle_ops():
        plain = net(x).asnumpy()
        assert qfuse.fuse_inference(net) == 27
        seen = []
        real = ops.qconv2d

        def spy(*a, **k):
            seen.append((k.get("in_stat") is not None, k.get("bn_scale") is not None, k.get("act")))
            return real(*a, **k)
        ops.qconv2d = spy
        try:
            fused = net(x).asnumpy()
        finally:
            ops.qconv2d = real
        assert len(seen) == 26 and all(s == (True, True, "relu") for s in seen)
        assert np.isfinite(fused).all() and np.abs(fused - plain).max() <= 0.1 * np.abs(plain).max() + 1e-3
        qfuse.unfuse(net)
        np.testing.assert_array_equal(net(x).asnumpy(), plain)


def test_collection_with_binning_producers_equals_one_pass_per_block():
    """KL collection of a fused net (round 4): from the second batch on the BatchNorm / residual passes bin what they store
    (quantize/fuse.py sinks, `ops.bn_act_stat(..., hist=)`); blocks fed by one producer share a range and a histogram.  Same
    histograms and ranges as with `FQ_KL_FUSED_HIST=0` (one pass per block and batch); the passes are counted.  On the oracle."""
    from quantization.mxnet_amd import ops
    from quantization.mxnet_amd.mx.gluon.model_zoo import get_model
    from quantization.mxnet_amd.quantize import distribution_calibrate as dc, fuse
    reset_naming()
    np.random.seed(3)
    net = get_model("cifar_resnet20_v1", classes=10)
    fn = {nn.Conv2D: convert.gen_conv2d_converter(quantize_input=True, quant_type="channel"),
          nn.Dense: convert.gen_dense_converter(quantize_input=True, quant_type="channel"), nn.Activation: None,
          nn.BatchNorm: None}
    # (the first unit's first convolution sees the un-activated head of the CIFAR ResNets: negative values, which a collection
    # refuses as the reference does)
    convert.convert_model(net, exclude=[net.features[0], net.features[1], net.features[2][0].body[0],
                                        net.features[2][0].body[1]], convert_fn=fn)
    net.initialize(mx.init.Xavier())
    qparams_init(net)
    rng = np.random.default_rng(9)
    loader = [(mx.nd.array(rng.standard_normal((2, 3, 16, 16)).astype(np.float32)), None) for _ in range(3)]
    with oracle_ops():
        net(loader[0][0])
        net.disable_quantize()
        fuse.fuse_inference(net)
        blocks = net.collect_quantized_blocks()
        res, passes = {}, {}
        real = ops.histogram_accumulate
        try:
            for fused in (True, False):
                n = [0]

                def counted(*a, **k):
                    n[0] += 1
                    return real(*a, **k)
                ops.histogram_accumulate = counted
                dc.FUSED_HISTOGRAMS = fused
                res[fused] = dc.collect_feature_maps(net, 2048, loader, mx.cpu())
                passes[fused] = n[0]
        finally:
            ops.histogram_accumulate = real
            dc.FUSED_HISTOGRAMS = True
        assert fuse._collection is None
        for b in blocks:
            assert res[True][1][b] == res[False][1][b], b.name
            np.testing.assert_array_equal(res[True][0][b], res[False][0][b], err_msg=b.name)
        # (the oracle's stand-in for a binning producer IS a histogram pass over its result: the count stays the same where a
        # producer took over and drops by the blocks that share a tensor with a sibling)
        assert passes[False] == 3 * len(blocks) and len(blocks) - 3 <= passes[True] / 3 <= len(blocks)
        assert passes[True] < passes[False]


def test_scalar_operand_cache_keeps_its_entries_and_tells_signed_zeros_apart():
    """the device-scalar cache behind `x * c` must never drop an entry (a captured hipGraph may hold its address)
    and must key on the bit pattern (x / -0.0 is -inf, x / 0.0 is +inf)."""
    from quantization.mxnet_amd.mx import ndarray as nd_mod
    x = mx.nd.array(np.float32([1.0, 2.0]))
    assert np.all(np.isposinf((x / 0.0).asnumpy())) and np.all(np.isneginf((x / -0.0).asnumpy()))
    assert np.array_equal(np.signbit((x * -0.0).asnumpy()), [True, True])
    first = nd_mod._operand(0.9, x._t)
    saved, nd_mod._SCALARS_MAX = nd_mod._SCALARS_MAX, len(nd_mod._scalars)      # the table is "full" from here on
    try:
        t = nd_mod._operand(123.456, x._t)
        assert float(t) == pytest.approx(123.456) and nd_mod._operand(0.9, x._t) is first
        assert nd_mod._operand(123.456, x._t) is not t                            # uncached, but right
        assert len(nd_mod._scalars) == nd_mod._SCALARS_MAX
    finally:
        nd_mod._SCALARS_MAX = saved


def test_subsampled_trunk_links_and_host_plumbing_on_the_oracle():
    """quantize/fuse.py `sub_next` + convert_conv2d.sub_target / _convolve (round 6), with the oracle standing in for the library:
    the closing 1x1 of the last unit of ResNet-50's stages 1-3 is linked to the two stride-2 1x1 readers of the next stage (none in
    ResNet-18, whose units open with a 3x3); with the link honoured the producer hands y[:, :, ::2, ::2] and the statistic of all
    of y, the readers run with stride 1 - same logits, same `current_input_max` everywhere; a reader that cannot run on the
    integer codes convolves the subsampled tensor with stride 1 (convert_conv2d.py:108 with the block's kwargs otherwise)."""
    from quantization.mxnet_amd import ops
    from quantization.mxnet_amd.mx.gluon.model_zoo import get_model
    from quantization.mxnet_amd.quantize import fuse

    def build(name):
        reset_naming()
        np.random.seed(4)
        net = get_model(name, classes=10)
        fn = {nn.Conv2D: convert.gen_conv2d_converter(quantize_input=True, quant_type="channel"),
              nn.Dense: convert.gen_dense_converter(quantize_input=True, quant_type="channel"), nn.Activation: None,
              nn.BatchNorm: None}
        convert.convert_model(net, exclude=[net.features[0], net.features[1]], convert_fn=fn)
        net.initialize(mx.init.Xavier())
        qparams_init(net)
        return net

    def links(net):
        found = []
        net.apply(lambda b: found.append(b._fq_pw_fused["sub_next"]) if getattr(b, "_fq_pw_fused", None) and
                  b._fq_pw_fused.get("sub_next") else None)
        return found
    x = mx.nd.array(np.random.default_rng(2).standard_normal((2, 3, 40, 40)).astype(np.float32))
    pool_was, fuse.STEM_POOL = fuse.STEM_POOL, False        # (the oracle's first convolution does not pool)
    try:
        _subsampled_trunk_checks(build, links, x, ops, fuse)
    finally:
        fuse.STEM_POOL = pool_was


def _subsampled_trunk_checks(build, links, x, ops, fuse):
    with oracle_ops():
        r18 = build("resnet18_v1")
        r18(x)
        fuse.fuse_inference(r18)
        assert links(r18) == []
        net = build("resnet50_v1")
        net.quantize_input(enable=True, online=True)
        net(x)
        net.fix_params()
        fuse.fuse_inference(net)
        found = links(net)
        assert len(found) == 3
        for lk in found:
            first, sc = lk["readers"]
            assert first is list(lk["unit"].body._children.values())[0] and sc is list(lk["unit"].downsample._children.values())[0]
            assert first._kwargs["stride"] == (2, 2) and sc._kwargs["stride"] == (2, 2)
        outs = {}
        real = ops.pwconv_i8
        for mode in ("whole", "sub", "sub, one reader off the codes"):
            seen = []

            def spy(*a, **k):
                if k.get("subsample"):
                    seen.append(tuple(a[0].shape))
                return real(*a, **k)
            ops.pwconv_i8 = spy
            old, fuse.SUBSAMPLE = fuse.SUBSAMPLE, mode != "whole"
            ops.StatArena._tls.current = object()        # (what the rewired net's forward sets on a GPU: "inside the net")
            if mode.endswith("codes"):
                found[1]["readers"][1]._fq_no_int8 = True
            try:
                out = net(x).asnumpy()
                cur = [float(b.current_input_max) for b in net.collect_quantized_blocks()]
            finally:
                ops.pwconv_i8 = real
                fuse.SUBSAMPLE = old
                ops.StatArena._tls.current = None
            outs[mode] = (out, cur, seen)
        assert outs["whole"][2] == [] and len(outs["sub"][2]) == 3 and len(outs["sub, one reader off the codes"][2]) == 3
        # (the same forwards fold four shortcut convolutions into their units' closing 1x1 - the oracle's stand-in composes the two
        # launches - and the reader that left the integer path has its record materialised)
        np.testing.assert_array_equal(outs["sub"][0], outs["whole"][0])
        assert outs["sub"][1] == outs["whole"][1]
        # (the fp32 library convolution of the fake-quantised tensors adds in another order than the integer path: close, not equal)
        np.testing.assert_allclose(outs["sub, one reader off the codes"][0], outs["whole"][0], rtol=0, atol=2e-3 * np.abs(outs["whole"][0]).max())
        assert outs["sub, one reader off the codes"][1][:5] == outs["whole"][1][:5]


def test_vgg16_through_convert_model_and_collect_qparams_as_the_reference_test_does():
    """The reference's tests/test_collect_qparams.py: `vgg16(pretrained=True)` -> `convert_model(net)` -> `qparams_init(net)` ->
    `collect_qparams(net)` / `print_all_qparams(net)` (it asserts nothing; here: one `input_max` per Conv2D and Dense, gluon's
    parameter names).  And the zoo's `batch_norm` argument, which the reference CLI passes to vgg only (simulate_quantization.py:197)."""
    from quantization.mxnet_amd.mx.gluon.model_zoo import get_model, get_model_list
    from quantization.mxnet_amd.quantize import collect_qparams, print_all_qparams
    np.random.seed(7)
    net = get_model("vgg16", pretrained=False)
    convert.convert_model(net)
    qparams_init(net)
    q = collect_qparams(net)
    assert len(q) == 16 and sorted(q)[0] == "vgg0_conv0_input_max" and "vgg0_dense2_input_max" in q      # 13 convolutions + 3 Dense
    print_all_qparams(net)
    assert {"vgg11", "vgg13", "vgg16", "vgg19", "vgg16_bn"} <= set(get_model_list())
    bn = get_model("vgg16", pretrained=False, batch_norm=True)
    assert sum(type(b) is nn.BatchNorm for b in bn.features._children.values()) == 13
    with pytest.raises(TypeError):
        get_model("mobilenet1.0", pretrained=False, batch_norm=True)
    x = mx.nd.array(np.random.default_rng(0).standard_normal((2, 3, 32, 32)).astype(np.float32))
    small = get_model("vgg11", pretrained=False, classes=10)
    assert small(x).shape == (2, 10)
    # several batches in flight only for nets without library GEMMs (quantize.fuse.library_gemm_blocks)
    from quantization.mxnet_amd.quantize import fuse
    assert len(fuse.library_gemm_blocks(small)) == 3
    mb = get_model("mobilenet1.0", pretrained=False, classes=10)
    assert len(fuse.library_gemm_blocks(mb)) == 1            # (not converted, not fused: its classifier is the library's)
    mb.output._fq_dense_int8 = True
    assert fuse.library_gemm_blocks(mb) == []


def test_dense_on_an_unflattened_input_takes_the_statistic_the_reference_takes():
    """convert_dense.py:41 - `F.max(F.abs(x), axis=1).mean()` - on an (N, C, H, W) input (vgg's first Dense, fed by a pooling
    layer) reduces over C only and averages N * H * W maxima: not the per-sample maximum a flattened input gives.  The
    reference's own expression through the facade, the oracle, and the converted block agree."""
    from oracle import fq_oracle as O
    rng = np.random.default_rng(12)
    x = rng.standard_normal((3, 8, 2, 2)).astype(np.float32)
    xa = mx.nd.array(x)
    want_cur = mx.nd.max(mx.nd.abs(xa), axis=1).mean().asscalar()                       # the reference's line, verbatim
    y, cur, _, _ = O.dense_input_fake_quant(x, False, 8)
    assert np.float32(cur) == np.float32(want_cur) and cur != O.batch_mean(O.absmax_per_sample(x))
    reset_naming()
    net = nn.HybridSequential()
    net.add(nn.Dense(4, in_units=32))
    net.initialize(mx.init.Xavier())
    convert.convert_model(net, convert_fn={nn.Dense: convert.gen_dense_converter(quantize_input=True)})
    qparams_init(net)
    blk = net[0]
    seen = {}
    real = blk.origin_forward
    blk.origin_forward = lambda F, xq, w, b=None: (seen.setdefault("xq", xq.asnumpy()), real(F, xq, w, b))[1]
    with oracle_ops():
        blk(xa)
    assert np.float32(float(blk.current_input_max)) == np.float32(want_cur)
    np.testing.assert_array_equal(seen["xq"], y)


def test_every_environment_variable_the_product_reads_is_listed_in_docs_knobs():
    """the tuning surface is one table - a variable read anywhere in the package, bench.py or the examples and
    missing from docs/knobs.md fails here."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [f for f in glob.glob(os.path.join(root, "quantization", "**", "*"), recursive=True)
             if f.endswith((".py", ".hip", ".h"))] + [os.path.join(root, "bench.py")] + glob.glob(os.path.join(root, "examples", "*.py"))
    names = set()
    for f in files:
        t = open(f, errors="ignore").read()
        names |= set(re.findall(r'(?:env_int|getenv|environ\.get|environ\[|setdefault)\(\s*["\'](FQ_[A-Z0-9_]+)["\']', t))
    doc = open(os.path.join(root, "docs", "knobs.md")).read()
    assert len(names) >= 70
    assert sorted(n for n in names if n not in doc) == []


def test_every_entry_point_of_the_header_is_named_in_integration_md():
    """The drop-in boundary is documented entry by entry: a function declared in include/fakequant.h and missing from
    INTEGRATION.md (what it replaces in the reference, or what helper it is) fails here."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "fakequant.h")).read()
    names = set(re.findall(r'\b(fq_[a-z0-9_]+)\s*\(', re.sub(r'/\*.*?\*/', '', header, flags=re.S)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert len(names) >= 80
    assert sorted(n for n in names if n not in doc) == []
