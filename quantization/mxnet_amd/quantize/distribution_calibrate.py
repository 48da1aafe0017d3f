"""`collect_feature_maps`, `kl_calibrate` — reference: quantize/distribution_calibrate.py:27-171.

Reference data flow: every quantised block's input is copied to the host (`x[0].asnumpy()`, :84) for
every calibration batch — 2.56 GB (MobileNet) / 5.11 GB (ResNet-50) of PCIe traffic per batch — then histogrammed by
single-threaded numpy, and the KL search is a Python double loop (~1.4 s per layer).

Here nothing leaves the GPU: the forward hook keeps a reference to the block's input tensor, `fq_global_max` fixes the
range on the first batch (:97-101), `fq_histogram_accumulate` adds exact uint64 counts in place (:103-104), and
`fq_kl_search` evaluates all candidate thresholds of all layers in one launch with the reference's arithmetic and
summation order.  Return types match the reference: `{block: np.float32 histogram[bins]}`, `{block: np.float32 max}`,
and `kl_calibrate -> int`.
"""
import numpy as np
import torch
from tqdm import tqdm

from ..mx.ndarray import NDArray
from .. import ops

__all__ = ['collect_feature_maps', 'kl_calibrate', 'kl_calibrate_many']


class _LayerHist(object):
    __slots__ = ("hist", "fm_max", "neg")

    def __init__(self, bins, device):
        self.hist = torch.zeros(bins, dtype=torch.int64, device=device)
        self.neg = torch.zeros(1, dtype=torch.int32, device=device)
        self.fm_max = None


def collect_feature_maps(net, bins, loader, ctx, tqdm_desc="Collect FM", sync=None):
    """
    Collect feature maps and record discrete histograms.
    :param net: converted net (has `collect_quantized_blocks`)
    :param bins: int, number of histogram bins
    :param loader: iterable of (X, y) batches
    :param ctx: context the batches are moved to
    :param tqdm_desc: str
    :param sync: optional multi-GPU hook `(stage, tensor) -> None` (dist.py): called with ("range", fm_max vector) after
        the first batch and ("hist", all histograms) at the end, so every rank ends with the global statistics.
    :return: (hist_collector, fm_max_collector) keyed by block, as in the reference.
    """
    quantized_blocks = net.collect_quantized_blocks()

    """ Add hooks to quantized blocks """
    hooks = []
    fm_collector = {}
    for blk in quantized_blocks:
        def _collect(m, x, y):
            # (a fused producer hands no int8 codes to a hooked block - convert_conv2d.handover_target - so x[0] is the fp32
            # activation whether or not the net quantises its inputs offline while it is being collected)
            assert getattr(x[0], "_fq_c16", None) is None, "collect_feature_maps: block input is an int8 code tensor"
            fm_collector.setdefault(m, []).append(x[0])          # device tensor reference, no copy (cf. :84)
        hooks.append(blk.register_forward_hook(_collect))

    """ Collect feature maps """
    state = {}
    range_shared = sync is None

    def share_range():
        # The FIRST batch fixes every layer's range (:97-101).  Under sharding that is global batch 0, which rank 0 holds:
        # its ranges are broadcast, so the counts every rank adds up are those one device would have produced.  A rank
        # whose shard is empty still takes part (with placeholders), so all ranks issue the same collectives.
        device = ctx.torch_device if hasattr(ctx, "torch_device") else None
        order = [m for m in quantized_blocks if m in state] or quantized_blocks
        mine = [state[m].fm_max if m in state else torch.zeros(1, dtype=torch.float32, device=device) for m in order]
        packed = torch.cat(mine)
        sync("range", packed)
        for i, m in enumerate(order):
            st = state.get(m)
            if st is None:
                st = state[m] = _LayerHist(bins, packed.device)
            st.fm_max = packed[i:i + 1].clone()

    with tqdm(total=len(loader), desc=tqdm_desc) as pbar:
        for X, _ in loader:
            X = X.as_in_context(ctx)
            _ = net(X)
            inputs = {}
            for m, fms in fm_collector.items():
                t = fms[0]._t if len(fms) == 1 else torch.cat([f._t for f in fms], dim=0)     # :94
                inputs[m] = t if t.is_contiguous() else t.contiguous()
                if m not in state:
                    st = state[m] = _LayerHist(bins, t.device)
                    st.fm_max = ops.global_max(inputs[m])         # first chunk sets the range (:97-101)
            if not range_shared:
                share_range()
                range_shared = True
            for m, t in inputs.items():
                st = state[m]
                ops.histogram_accumulate(t, st.fm_max, st.hist, st.neg)       # :39-45 and :103-104
            fm_collector.clear()
            pbar.update(1)
    if not range_shared:                      # this rank's shard was empty
        share_range()

    """ Delete hooks """
    for h in hooks:
        h.detach()

    if sync is not None:
        order = [m for m in quantized_blocks if m in state]
        packed = torch.stack([state[m].hist for m in order])
        sync("hist", packed)
        for i, m in enumerate(order):
            state[m].hist = packed[i]

    # one synchronisation for the whole calibration: checks the reference asserted per batch (:35-36), then results
    hist_collector, fm_max_collector = {}, {}
    for m, st in state.items():
        assert int(st.neg.item()) == 0, "Activation should >=0"
        mx_ = np.float32(st.fm_max.item())
        assert mx_ > 0, "Bad distribution: all zero-value"
        hist_collector[m] = ops.hist_to_float(st.hist).cpu().numpy()
        fm_max_collector[m] = mx_
    return hist_collector, fm_max_collector


def kl_calibrate_many(hists, levels, min_bins, bins, device=None):
    """`kl_calibrate` for a list of histograms in ONE launch; returns a list of ints."""
    assert min_bins >= levels, f"min_bins should be greater than levels ({min_bins} vs. {levels})"
    arr = np.stack([np.asarray(h, dtype=np.float32).reshape(-1) for h in hists])
    assert arr.shape[1] == bins or arr.shape[1] > min_bins, "histogram length does not match `bins`"
    if device is None:
        device = ops.default_device("kl_calibrate")
    t = torch.from_numpy(np.ascontiguousarray(arr[:, :bins] if arr.shape[1] >= bins else arr)).to(device)
    best = ops.kl_search(t, levels, min_bins)
    return [int(b) for b in best.cpu().numpy()]


def kl_calibrate(data, levels, min_bins, bins):
    """
    KL-divergence calibration for offline-quantization (same contract as the reference, :117-171).
    :param data: numpy.ndarray, discrete histogram for activation data.
    :param levels: int, number of levels to quantize into.
    :param min_bins: int, minimal number of bins to search (should be >= levels).
    :param bins: int, maximal number of bins to search.
    :return: int, best number of bins (the caller turns it into a threshold: (best + 0.5) * fm_max / bins).
    """
    if isinstance(data, NDArray):
        data = data.asnumpy()
    return kl_calibrate_many([data], levels, min_bins, bins)[0]
