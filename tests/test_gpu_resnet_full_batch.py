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
