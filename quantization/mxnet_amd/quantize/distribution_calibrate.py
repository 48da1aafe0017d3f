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
import os

import numpy as np
import torch
from tqdm import tqdm

from ..mx.ndarray import NDArray
from .. import ops

__all__ = ['collect_feature_maps', 'kl_calibrate', 'kl_calibrate_many']


FUSED_HISTOGRAMS = os.environ.get("FQ_KL_FUSED_HIST", "1") != "0"     # (0: every histogram in a pass of its own, as round 3)


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
    from . import fuse as _fuse
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
    # Fused producers (quantize/fuse.py) tag what they make with (producer, sink): blocks fed by the same producer see the
    # same tensor every batch - one range, one histogram (they SHARE a _LayerHist) - and from the second batch on the
    # producer's own pass adds the counts (ops.bn_act_stat / add_act_stat with `hist=`), so no pass here re-reads the tensor.
    sinks = _fuse.begin_collection() if FUSED_HISTOGRAMS and bins <= ops.HIST_FUSED_MAX_BINS else None
    by_producer, ambiguous = {}, set()

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

    try:
        with tqdm(total=len(loader), desc=tqdm_desc) as pbar:
            for X, _ in loader:
                X = X.as_in_context(ctx)
                if sinks is not None:
                    _fuse.collection_calls().clear()
                _ = net(X)
                pending = {}          # id(histogram state) -> (state, tensor): ONE separate pass per histogram and batch
                seen = {}             # producer -> the NDArray it made in this forward
                if sinks:
                    # a producer that binned its output into a shared histogram: a block that reads it must have run too
                    # (a net whose forward skips blocks from batch to batch cannot hand its histograms to the producers)
                    fed = {id(state[m]) for m in fm_collector if m in state}
                    for p_, st_ in sinks.items():
                        if id(st_) not in fed:
                            raise RuntimeError("collect_feature_maps: a producer binned a tensor no collected block read in "
                                               "this forward (FQ_KL_FUSED_HIST=0 collects such nets)")
                for m, fms in fm_collector.items():
                    tag = getattr(fms[0], "_fq_kl", None) if len(fms) == 1 and sinks is not None else None
                    if tag is not None and seen.setdefault(tag[0], fms[0]) is not fms[0]:
                        ambiguous.add(tag[0])                     # one producer, two tensors in a forward: never share
                        tag = None
                    st = state.get(m)
                    if st is None and tag is not None and tag[0] in by_producer:
                        st = state[m] = by_producer[tag[0]]       # a sibling consumer of the same tensor: one range, one histogram
                    if st is not None and tag is not None and tag[1] is st:
                        continue                                  # the producer binned this tensor while it stored it
                    if st is not None and id(st) in pending:
                        continue                                  # the sibling's pass below covers the shared histogram
                    t = fms[0]._t if len(fms) == 1 else torch.cat([f._t for f in fms], dim=0)     # :94
                    t = t if t.is_contiguous() else t.contiguous()
                    if st is None:
                        st = state[m] = _LayerHist(bins, t.device)
                        st.fm_max = ops.global_max(t)             # first chunk sets the range (:97-101)
                        if tag is not None:
                            by_producer[tag[0]] = st
                    pending[id(st)] = (st, t)
                if not range_shared:
                    share_range()
                    range_shared = True
                for st, t in pending.values():
                    ops.histogram_accumulate(t, st.fm_max, st.hist, st.neg)       # :39-45 and :103-104
                if sinks is not None:
                    # ranges are fixed: producers that ran exactly once in this forward take over from the next batch
                    calls = _fuse.collection_calls()
                    for p in list(sinks):
                        if calls.get(p, 0) != 1:
                            raise RuntimeError("collect_feature_maps: a producer that bins its output ran %d times in one "
                                               "forward (the net changed during the collection)" % calls.get(p, 0))
                    sinks.update({p: st for p, st in by_producer.items() if calls.get(p, 0) == 1 and p not in ambiguous})
                fm_collector.clear()
                pbar.update(1)
    finally:
        if sinks is not None:
            _fuse.end_collection()
        """ Delete hooks """
        for h in hooks:
            h.detach()
    if not range_shared:                      # this rank's shard was empty
        share_range()

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
