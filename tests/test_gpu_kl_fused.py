"""KL collection with the histogram in the producer's pass (round 4; fq_bn_act_stat_hist / fq_add_act_stat_hist): while
`collect_feature_maps` (reference quantize/distribution_calibrate.py:91-106) gathers a block's input histogram, the fused
BatchNorm / residual pass that stores the tensor also bins it from the second batch on.  Counts are integers: everything here
is compared for EQUALITY - with the separate pass (fq_histogram_accumulate), with the host twin, and net against net."""
import numpy as np
import pytest
import torch

from oracle import host as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(gpu):
    return gpu.torch_device


@pytest.fixture(scope="module")
def ops():
    from quantization.mxnet_amd import ops as _ops
    return _ops


class Sink(object):
    def __init__(self, bins, mx_, dev):
        self.fm_max = torch.tensor([mx_], dtype=torch.float32, device=dev)
        self.hist = torch.zeros(bins, dtype=torch.int64, device=dev)
        self.neg = torch.zeros(1, dtype=torch.int32, device=dev)


# (n, c, h, w): 16-byte rows and ragged ones, planes of 49, more chunks than the grid holds, a single small chunk
SHAPES = [(128, 64, 56, 56), (7, 33, 7, 7), (3, 5, 9, 11), (64, 256, 14, 14), (1, 8, 4, 4), (16, 1024, 7, 7)]


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
@pytest.mark.parametrize("act", ["relu", "relu6", "none"])
@pytest.mark.parametrize("bins", [2048, 128])
def test_batchnorm_pass_bins_what_it_stores(dev, ops, shape, act, bins):
    rng = np.random.default_rng(hash((shape, act, bins)) % 2 ** 31)
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * 2).to(dev)
    sc = torch.from_numpy((rng.random(shape[1]) + 0.5).astype(np.float32)).to(dev)
    sh = torch.from_numpy(rng.standard_normal(shape[1]).astype(np.float32)).to(dev)
    y0, s0 = ops.bn_act_stat(x, sc, sh, act)
    mx_ = float(y0.max()) * 0.8                       # a range the tensor exceeds: the clip of :39 is exercised
    want, got = Sink(bins, mx_, dev), Sink(bins, mx_, dev)
    for _ in range(2):                                # accumulates, as over batches
        ops.histogram_accumulate(y0, want.fm_max, want.hist, want.neg)
        y1, s1 = ops.bn_act_stat(x, sc, sh, act, hist=got)
        assert torch.equal(y1, y0) and torch.equal(s1, s0)
    assert torch.equal(got.hist, want.hist) and int(got.neg) == int(want.neg)
    assert (act != "none") == (int(got.neg) == 0)
    yh, sh_, h, neg = H.bn_act_hist(x.cpu().numpy(), sc.cpu().numpy(), sh.cpu().numpy(), act, mx_, bins)      # the host twin
    assert np.array_equal(yh, y0.cpu().numpy()) and np.array_equal(sh_, s0.cpu().numpy())
    assert np.array_equal(got.hist.cpu().numpy(), 2 * h.astype(np.int64)) and int(got.neg) == 2 * int(neg)


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
@pytest.mark.parametrize("bins", [2048, 4096])
def test_residual_pass_bins_what_it_stores(dev, ops, shape, bins):
    rng = np.random.default_rng(hash((shape, bins)) % 2 ** 31)
    a = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dev)
    y0, s0 = ops.add_act_stat(a, b, "relu")
    mx_ = float(y0.max())
    want, got = Sink(bins, mx_, dev), Sink(bins, mx_, dev)
    ops.histogram_accumulate(y0, want.fm_max, want.hist, want.neg)
    y1, s1 = ops.add_act_stat(a, b, "relu", hist=got)
    assert torch.equal(y1, y0) and torch.equal(s1, s0)
    assert torch.equal(got.hist, want.hist) and int(got.neg) == 0 and int(got.hist.sum()) == int((y0 > 0).sum())
    yh, _, h, _ = H.add_act_hist(a.cpu().numpy(), b.cpu().numpy(), "relu", mx_, bins)                            # the host twin
    assert np.array_equal(yh, y0.cpu().numpy()) and np.array_equal(got.hist.cpu().numpy(), h.astype(np.int64))


def test_fused_form_refuses_what_it_cannot_hold(dev, ops):
    x = torch.zeros(2, 4, 8, 8, device=dev)
    one = torch.ones(4, device=dev)
    with pytest.raises(ValueError):
        ops.bn_act_stat(x, one, one, "relu", hist=Sink(8192, 1.0, dev))
    with pytest.raises(ValueError):
        ops.bn_act_stat(x, one, one, "relu", want_stat=False, hist=Sink(2048, 1.0, dev))


def _collect(net, loader, gpu, fused, counts=None):
    from quantization.mxnet_amd import ops
    from quantization.mxnet_amd.quantize import distribution_calibrate as dc
    real = ops.histogram_accumulate, ops.global_max, dc.FUSED_HISTOGRAMS
    if counts is not None:
        def hist(x, *a, **k):
            counts["hist"] += 1
            return real[0](x, *a, **k)

        def gmax(x):
            counts["max"] += 1
            return real[1](x)
        ops.histogram_accumulate, ops.global_max = hist, gmax
    dc.FUSED_HISTOGRAMS = fused
    try:
        return dc.collect_feature_maps(net, 2048, loader, gpu)
    finally:
        ops.histogram_accumulate, ops.global_max, dc.FUSED_HISTOGRAMS = real


# (MobileNetV2's linear bottlenecks feed negative values to quantised blocks: the reference's collection asserts on them, :35)
@pytest.mark.parametrize("model,hw,separate_after_first", [("resnet50_v1", 64, 3), ("mobilenet1.0", 64, 15),
                                                           ("resnet18_v1", 32, 2)])
def test_collection_of_a_fused_net_equals_the_separate_passes(gpu, model, hw, separate_after_first):
    """Config 3's collection on fused nets: histograms and ranges with the producers binning (`FQ_KL_FUSED_HIST`, default on)
    == with one pass per block (round 3) == the host twin on the captured block inputs; and the passes that remain are counted:
    the first batch takes one range + one histogram pass per distinct TENSOR (blocks fed by the same producer share them: 50 for
    ResNet-50's 53 blocks), later batches only what no BatchNorm / residual pass makes - ResNet-50: 3 of 50 (the pooled head
    that feeds the first unit, the classifier's input, ...); MobileNet: the 15 tensors the depthwise kernel and the first
    convolution store (their epilogues do not bin)."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build
    net = _build(model, 1000, gpu, quant_type="channel")
    rng = np.random.default_rng(11)
    loader = [(mx.nd.array(rng.standard_normal((4, 3, hw, hw)).astype(np.float32), ctx=gpu), None) for _ in range(3)]
    net.disable_quantize()
    fuse.fuse_inference(net)
    blocks = net.collect_quantized_blocks()
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True        # (the fp32 convolutions of two runs must agree bit for bit to compare them)
    try:
        _collection_checks(net, blocks, loader, gpu, model, separate_after_first)
    finally:
        torch.backends.cudnn.deterministic = det


def _collection_checks(net, blocks, loader, gpu, model, separate_after_first):
    from quantization.mxnet_amd.quantize import fuse
    want_hist, want_max = {}, {}

    def watch(m, args):
        fm = args[0]._t.detach().cpu().numpy()
        if m not in want_max:
            want_max[m] = H.global_max(fm)
            want_hist[m] = np.zeros(2048, np.uint64)
        H.histogram_accumulate(fm, want_max[m], 2048, want_hist[m])
    hooks = [b.register_forward_pre_hook(watch) for b in blocks]
    c1 = {"hist": 0, "max": 0}
    h1, r1 = _collect(net, loader, gpu, True, c1)
    for h in hooks:
        h.detach()
    c0 = {"hist": 0, "max": 0}
    h0, r0 = _collect(net, loader, gpu, False, c0)
    assert set(h1) == set(h0) == set(blocks)
    for b in blocks:                                  # the run that was watched: against the host twin on the same tensors
        assert r1[b] == want_max[b], b.name
        assert np.array_equal(h1[b], want_hist[b].astype(np.float32)), b.name
    for b in blocks:
        assert r1[b] == r0[b] and np.array_equal(h1[b], h0[b]), b.name
    assert c0 == {"hist": 3 * len(blocks), "max": len(blocks)}
    assert c1["max"] <= len(blocks) and c1["hist"] < c0["hist"], (c1, c0)
    if separate_after_first is not None:
        assert c1["hist"] == c1["max"] + 2 * separate_after_first, (c1, separate_after_first)
    # a second collection starts from scratch (no sink survives the first), and the net still runs outside a collection
    h2, r2 = _collect(net, loader, gpu, True)
    assert all(np.array_equal(h2[b], h1[b]) and r2[b] == r1[b] for b in blocks)
    assert fuse._collection is None
    assert np.isfinite(net(loader[0][0]).asnumpy()).all()
