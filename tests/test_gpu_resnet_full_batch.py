"""Round 6's ResNet forms at the BENCHMARK'S batch (128): the small-batch parity cases of test_gpu_shortcut.py / test_gpu_sub2.py do not
reach what only a full batch selects - eight-wavefront workgroups, the cache policy of outputs beyond 150 MB (nontemporal stores),
grids of several workgroups per CU.  The same checks (fused launch against the launches it replaces and against the host twin, bit
for bit) on the four stage shapes of ResNet-50 at batch 128, then the whole net at (128, 3, 224, 224) with the round's three switches
on against the same net with them off, and twice in a row.
Reference: gluon model_zoo BottleneckV1 (`(body(x) + downsample(x)).relu()`), its 1x1 Conv2D blocks wrapped by
quantize/convert/convert_conv2d.py:53-66,108."""
import numpy as np
import pytest
import torch

import test_gpu_shortcut as S
import test_gpu_sub2 as B

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


# (n, cin, cin2, cout, h, w): a stage's first unit - closing 1x1 + the unit's shortcut 1x1 (stages 2-4 read the subsampled trunk)
SHORT = [(128, 64, 64, 256, 56, 56), (128, 128, 256, 512, 28, 28), (128, 256, 512, 1024, 14, 14), (128, 512, 1024, 2048, 7, 7)]


# (the c16 form below is what offline evaluation runs: one stored-threshold case of the fp32 form is enough here)
SHORT_MODES = [(c, "online_u8_relu") for c in SHORT] + [(SHORT[0], "offline_u8_relu")]


@pytest.mark.parametrize("case,mode", SHORT_MODES, ids=["%dx%d+%d->%d@%dx%d" % c + "-" + m.split("_")[0] for c, m in SHORT_MODES])
def test_folded_shortcut_at_batch_128(dev, ops, case, mode):
    S.test_folded_shortcut_equals_the_two_launches_and_the_host_twin(dev, ops, case, mode)
    k = S._make(case, mode)
    first = S._run(k, dev, ops, True)
    for _ in range(2):
        for a, b in zip(S._run(k, dev, ops, True), first):
            assert np.array_equal(a, b), "repeats of the folded launch differ"


@pytest.mark.parametrize("case", [c + (c[1] != 64,) for c in SHORT], ids=["%dx%d+%d->%d@%dx%d" % c for c in SHORT])   # (+ shortcut input as codes)
def test_folded_shortcut_on_codes_at_batch_128(dev, ops, case):
    S.test_folded_shortcut_under_stored_thresholds_equals_the_two_launches(dev, ops, case)


# (n, cin, cout, h, w): the last unit of stages 1-3 - its output has two readers, both 1x1 with stride 2
SUB = [(128, 64, 256, 56, 56), (128, 128, 512, 28, 28), (128, 256, 1024, 14, 14)]


@pytest.mark.parametrize("case", SUB, ids=["%dx%d->%d@%dx%d" % c for c in SUB])
def test_subsampled_boundary_at_batch_128(dev, ops, case):
    B.test_sub2_stores_the_even_pixels_of_the_whole_launch_and_keeps_its_statistic(dev, ops, case, "online_u8_bn_res_relu")


@pytest.mark.parametrize("case", SUB, ids=["%dx%d->%d@%dx%d" % c for c in SUB])
def test_subsampled_boundary_on_codes_at_batch_128(dev, ops, case):
    B.test_dual_sub2_stores_both_outputs_subsampled(dev, ops, case)


@pytest.mark.parametrize("config", ["online", "offline", "F43"])
def test_resnet50_at_batch_128_with_the_rounds_forms_equals_the_same_net_without(dev, ops, config):
    """BASELINE configurations 3 and 5 and ResNet-50 online as bench.py runs them: subsampled stage boundaries, folded shortcuts
    and the pooled last 1x1 on, against the same weights with the three switched off - logits and every current_input_max bit-equal,
    and a second forward of the same batch bit-equal to the first."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    g = torch.Generator(device=dev).manual_seed(77)
    X = mx.nd.NDArray(torch.randn(128, 3, 224, 224, device=dev, generator=g))
    small = [mx.nd.NDArray(X._t[4 * i:4 * i + 4].contiguous()) for i in range(3)]
    outs = {}
    names = ("SUBSAMPLE", "SHORTCUT_FUSE", "GAP_FUSE")
    try:
        for on in (False, True):
            net = _build("resnet50_v1", 1000, mx.gpu(0), quant_type="channel", wino="F43" if config == "F43" else "none")
            net.quantize_input(enable=True, online=True)
            if config == "offline":
                for x in small[:2]:
                    net(x)
                    net.update_ema()
            net.fix_params()
            net.quantize_input(enable=True, online=config != "offline")
            net(small[2])
            fuse.fuse_inference(net)
            old = [getattr(fuse, n) for n in names]
            for n in names:
                setattr(fuse, n, on)
            seen = []
            real = ops.pwconv_i8_shortcut
            ops.pwconv_i8_shortcut = lambda *a, **k: (seen.append(1), real(*a, **k))[1]
            try:
                out = net(X)._t.clone()
                cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
                again = net(X)._t.clone()
            finally:
                for n, v in zip(names, old):
                    setattr(fuse, n, v)
                ops.pwconv_i8_shortcut = real
            assert torch.equal(out, again), "two forwards of one batch differ (switches %s)" % on
            assert bool(torch.isfinite(out).all())
            outs[on] = (out.cpu().numpy(), cur, len(seen))
    finally:
        torch.backends.cudnn.deterministic = was
    assert outs[False][2] == 0 and outs[True][2] == 8, (outs[False][2], outs[True][2])        # four stage heads, two forwards
    S._eq(outs[True][0], outs[False][0], "logits")
    S._eq(outs[True][1], outs[False][1], "current_input_max of every block")


def _on_off(dev, model, kw, offline, names):
    """logits and every current_input_max of `model` at (128, 3, 224, 224), the fuse.py switches `names` off and on (a net per setting, the same
    seeded weights), each forward run twice"""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build
    g = torch.Generator(device=dev).manual_seed(5)
    X = mx.nd.NDArray(torch.randn(128, 3, 224, 224, device=dev, generator=g))
    small = [mx.nd.NDArray(X._t[4 * i:4 * i + 4].contiguous()) for i in range(3)]
    outs = {}
    for on in (False, True):
        net = _build(model, 1000, mx.gpu(0), **kw)
        net.quantize_input(enable=True, online=True)
        if offline:
            for x in small[:2]:
                net(x)
                net.update_ema()
        net.fix_params()
        net.quantize_input(enable=True, online=not offline)
        net(small[2])
        fuse.fuse_inference(net)
        old = [getattr(fuse, n) for n in names]
        for n in names:
            setattr(fuse, n, on)
        try:
            out = net(X)._t.clone()
            cur = np.asarray([float(b.current_input_max) for b in net.collect_quantized_blocks()], np.float32)
            again = net(X)._t.clone()
        finally:
            for n, v in zip(names, old):
                setattr(fuse, n, v)
        assert torch.equal(out, again), "two forwards of one batch differ (%s %s)" % (names, on)
        assert bool(torch.isfinite(out).all())
        outs[on] = (out.cpu().numpy(), cur)
    return outs


@pytest.mark.parametrize("model,kw", [("resnet50_v1", dict(quant_type="channel")), ("mobilenetv2_1.0", dict(quant_type="channel", wt=4))],
                         ids=["resnet50_v1", "mobilenetv2_1.0-w4"])
def test_offline_nets_at_batch_128_with_hand_overs_equal_the_same_nets_without(dev, ops, model, kw):
    """BASELINE configurations 3 and 4 as bench.py runs them: int8 codes between the fused layers (the wide code-to-code streaming
    forms, the dual closing 1x1 and the 3x3 on codes with eight wavefronts, the depthwise layer on codes - forms a full batch selects)
    against fp32 between the same layers."""
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    calls = []
    real = ops.pwconv_i8
    ops.pwconv_i8 = lambda x, *a, **k: (calls.append(isinstance(x, ops.Codes16)), real(x, *a, **k))[1]
    try:
        outs = _on_off(dev, model, kw, True, ("HANDOVER",))
    finally:
        ops.pwconv_i8 = real
        torch.backends.cudnn.deterministic = was
    assert sum(calls) >= 30, sum(calls)                      # (the switched-on forwards did read codes)
    S._eq(outs[True][0], outs[False][0], "logits")
    S._eq(outs[True][1], outs[False][1], "current_input_max of every block")


def test_mobilenet_at_batch_128_with_recompute_pairs_and_pooled_producer_equals_the_same_net_without(dev, ops):
    """The default workload as bench.py runs it (three recompute pairs, the last 1x1 storing the pooled means) against the same net
    with one storing launch per layer and the pooling pass."""
    seen = []
    real = ops.pwdw_fused
    ops.pwdw_fused = lambda *a, **k: (seen.append(1), real(*a, **k))[1]
    try:
        outs = _on_off(dev, "mobilenet1.0", dict(quant_type="layer"), False, ("RECOMPUTE", "GAP_FUSE"))
    finally:
        ops.pwdw_fused = real
    assert len(seen) == 6, len(seen)                         # three pairs, two forwards
    S._eq(outs[True][0], outs[False][0], "logits")
    S._eq(outs[True][1], outs[False][1], "current_input_max of every block")


def _np(t):
    return t.detach().cpu().numpy()


# (n, c, h, w): the 3x3 of every stage (c -> c)
C3 = [(128, 64, 56, 56), (128, 128, 28, 28), (128, 256, 14, 14), (128, 512, 7, 7)]


@pytest.mark.parametrize("case", C3, ids=["%dx%d@%dx%d" % c for c in C3])
@pytest.mark.parametrize("sliced", [False, True], ids=["one-slice", "three-slices-F43"])
def test_dense3x3_at_batch_128_against_the_host_twin(dev, ops, case, sliced):
    """fq_conv3x3_i8 / fq_conv3x3_i8_sliced on ResNet-50's four 3x3 shapes at the benchmark's batch (the eight-wavefront
    instantiations, 1.5 workgroups per CU on the 14x14 planes) against the C++ twin fed with the same tensors: output, per-sample
    statistic, current_input_max; and twice in a row."""
    from oracle import host as H
    from oracle import fq_oracle as O
    n, c, h, w = case
    g = torch.Generator(device=dev).manual_seed(c + h)
    x = torch.relu(torch.randn(n, c, h, w, device=dev, generator=g) * 2)
    wt = torch.randn(c, c, 3, 3, device=dev, generator=g) * (torch.rand(c, 1, 1, 1, device=dev, generator=g) * 0.2 + 0.02)
    sc = torch.rand(c, device=dev, generator=g) + 0.4
    sh = torch.randn(c, device=dev, generator=g)
    if sliced:
        wt = torch.from_numpy(O.wino_weight_fake_quant(_np(wt), "F43", 8)[0]).to(dev)      # the filter configuration 5 multiplies
        codes = ops.weight_slices_3x3(wt)
    else:
        codes = ops.weight_codes_3x3(wt, 1, 8)
    stat = ops.absmax_per_sample(x)
    cur = torch.zeros(1, device=dev)
    kw = dict(in_stat=stat, width=8, flags=0, bn_scale=sc, bn_shift=sh, act="relu")
    y, ystat = ops.conv3x3_i8(x, *codes, cur_out=cur, **kw)
    y2, ystat2 = ops.conv3x3_i8(x, *codes, cur_out=torch.zeros(1, device=dev), **kw)
    assert torch.equal(y, y2) and torch.equal(ystat, ystat2), "repeats differ"
    twin = H.conv3x3_i8_sliced if sliced else (lambda *a, **k: H.conv3x3_i8(a[0], a[1], 1, 8, **k))
    want, wstat = twin(_np(x), _np(wt), in_stat=_np(stat), bn_scale=_np(sc), bn_shift=_np(sh), act="relu", want_stat=True)
    S._eq(_np(y), want, "output")
    S._eq(_np(ystat), wstat, "statistic")
    assert float(cur) == float(H.batch_mean(_np(stat)))


# (n, cin, cout, h, w, stride, residual): first 1x1 of a unit (wide -> narrow), of a stage's first unit (stride 2, round 5 and before:
# now stride 1 on the subsampled trunk), closing 1x1 with the trunk as residual
PW = [(128, 256, 64, 56, 56, 1, False), (128, 512, 128, 28, 28, 1, False), (128, 1024, 256, 14, 14, 1, False),
      (128, 2048, 512, 7, 7, 1, False), (128, 256, 128, 56, 56, 2, False), (128, 64, 256, 56, 56, 1, True),
      (128, 128, 512, 28, 28, 1, True), (128, 256, 1024, 14, 14, 1, True), (128, 512, 2048, 7, 7, 1, True)]


@pytest.mark.parametrize("case", PW, ids=["%dx%d->%d@%dx%d-s%d%s" % (c[:6] + ("-res" if c[6] else "",)) for c in PW])
def test_pointwise_at_batch_128_against_the_host_twin(dev, ops, case):
    """fq_pwconv_i8 / fq_pwconv_i8_strided on ResNet-50's 1x1 shapes at the benchmark's batch - whatever form the shape-based choice
    takes there (stream / sample with the residual operand prefetched / split) - against the C++ twin; and twice in a row."""
    from oracle import host as H
    n, cin, cout, h, w, stride, res = case
    g = torch.Generator(device=dev).manual_seed(cin + cout + h)
    x = torch.relu(torch.randn(n, cin, h, w, device=dev, generator=g) * 1.7)
    wt = torch.randn(cout, cin, 1, 1, device=dev, generator=g) * 0.1
    sc = (torch.rand(cout, device=dev, generator=g) + 0.4) * torch.where(torch.rand(cout, device=dev, generator=g) < 0.1, -1.0, 1.0)
    sh = torch.randn(cout, device=dev, generator=g) * 0.3
    r = torch.randn(n, cout, h, w, device=dev, generator=g) * 2 if res else None
    codes = ops.weight_codes(wt, 1, 8)
    stat = ops.absmax_per_sample(x)
    kw = dict(in_stat=stat, width=8, flags=0, bn_scale=sc, bn_shift=sh, act="relu", residual=r)
    if stride != 1:
        kw["stride"] = stride
    cur = torch.zeros(1, device=dev)
    y, ystat = ops.pwconv_i8(x, *codes, None, cur_out=cur, **kw)
    y2, ystat2 = ops.pwconv_i8(x, *codes, None, cur_out=torch.zeros(1, device=dev), **kw)
    assert torch.equal(y, y2) and torch.equal(ystat, ystat2), "repeats differ"
    want, wstat = H.pwconv_i8(_np(x), _np(wt), 1, 8, in_stat=_np(stat), bn_scale=_np(sc), bn_shift=_np(sh), act="relu",
                              want_stat=True, stride=stride, residual=None if r is None else _np(r))
    S._eq(_np(y), want, "output")
    S._eq(_np(ystat), wstat, "statistic")
    assert float(cur) == float(H.batch_mean(_np(stat)))
