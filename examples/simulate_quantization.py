#!/usr/bin/env python3
"""Simulated-quantisation CLI for MI355X: evaluates a model-zoo network with fake-quantised weights and activations
through libfakequant, optionally after calibrating the activation ranges (naive EMA or KL).

The command line — every flag, default and help text in REFERENCE_FLAGS below — is the interface of the reference's
examples/simulate_quantization.py (hey-yahei/Quantization.MXNet, MIT licence, (c) YaHei; flags :49-103), kept so that the
reference's commands and scripts run unchanged.  The program behind it is this project's own:

    python examples/simulate_quantization.py --model=mobilenet1.0 --use-gpu=0
    python examples/simulate_quantization.py --model=resnet50_v1 --quant-type=channel --quantize-input-offline \
           --calib-mode=kl --use-gpu=0
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/simulate_quantization.py \
           --model=mobilenetv2_1.0 --quant-type=channel --weight-bits-width=4 --quantize-input-offline   # 8 x MI355X

What happens, in the reference's order (its __main__, :178-353): build the net -> `convert_model` with the converters the
flags describe, minus the excluded blocks -> `qparams_init` -> [calibrate: KL histograms + threshold search, or EMA of
the online statistic over `--calib-epoch` passes of a class-balanced sample of the training set] -> freeze the weights
-> evaluate top-1 / class-averaged accuracy.  Deliberate differences: no `.asscalar()` per layer, accuracy
counters and KL histograms / search on the device, optional one-process-per-GPU sharding (`torchrun`), synthetic datasets
and seeded weights when ImageNet / gluoncv checkpoints are absent (there is no network here).
"""
import argparse
import contextlib
import os
import sys
import time

import numpy as np
from tqdm import tqdm

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from quantization.mxnet_amd import mx, ops, dist as fqdist  # noqa: E402
from quantization.mxnet_amd.mx import cpu, gpu, nd  # noqa: E402
from quantization.mxnet_amd.mx.gluon import nn  # noqa: E402
from quantization.mxnet_amd.mx.gluon.data import Sampler, DataLoader, vision  # noqa: E402
from quantization.mxnet_amd.mx.gluon.model_zoo import get_model, get_model_list  # noqa: E402
from quantization.mxnet_amd.quantize import convert  # noqa: E402
from quantization.mxnet_amd.quantize.initialize import qparams_init  # noqa: E402
from quantization.mxnet_amd.quantize.distribution_calibrate import kl_calibrate_many, collect_feature_maps  # noqa: E402

# ---- the command line ------------------------------------------------------------------------------------------------------
# (flags, argparse keywords).  REFERENCE_FLAGS is the reference's interface, entry for entry; EXTRA_FLAGS are additions.
REFERENCE_FLAGS = [
    (['--model'], dict(type=str, default=None, help='type of model to use. see vision_model for options. (required)')),
    (['--print-model'], dict(action='store_true', help='print the architecture of model.')),
    (['--list-models'], dict(action='store_true', help='list all models supported for --model.')),
    (['--use-gpu'], dict(type=int, default=-1,
                         help='run model on gpu. (default: cpu — which this build refuses: HIP device required)')),
    (['--dataset'], dict(type=str, default="imagenet", choices=['imagenet', 'cifar10'],
                         help='dataset to evaluate (default: imagenet)')),
    (['--use-gn'], dict(action='store_true', help='whether to use group norm.')),
    (['--batch-norm'], dict(action='store_true', help='enable batch normalization or not in vgg. default is false.')),
    (['--use-se'], dict(action='store_true', help='use SE layers or not in resnext. default is false.')),
    (['--last-gamma'], dict(action='store_true',
                            help='whether to init gamma of the last BN layer in each bottleneck to 0.')),
    (['--merge-bn'], dict(action='store_true', help='merge batchnorm into convolution or not. (default: False)')),
    (['--weight-bits-width'], dict(type=int, default=8, help='bits width of weight to quantize into.')),
    (['--input-signed'], dict(type=str, default="false",
                              help='quantize inputs into int(true) or uint(fasle). (default: false)')),
    (['--input-bits-width'], dict(type=int, default=8, help='bits width of input to quantize into.')),
    (['--quant-type'], dict(type=str, default="layer", choices=['layer', 'group', 'channel'],
                            help='quantize weights on layer/group/channel. (default: layer)')),
    (['-j', '--num-data-workers'], dict(dest='num_workers', default=4, type=int,
                                        help='number of preprocessing workers (default: 4)')),
    (['--batch-size'], dict(type=int, default=128, help='evaluate batch size per device (CPU/GPU). (default: 128)')),
    (['--num-sample'], dict(type=int, default=5, help='number of samples for every class in trainset. (default: 5)')),
    (['--quantize-input-offline'], dict(action='store_true',
                                        help='calibrate via EMA on trainset and quantize input offline.')),
    (['--calib-mode'], dict(type=str, default="naive", choices=['naive', 'kl'],
                            help='how to calibrate inputs. (default: naive)')),
    (['--calib-epoch'], dict(type=int, default=3,
                             help='number of epoches to calibrate via EMA on trainset. (default: 3)')),
    (['--disable-cudnn-autotune'], dict(action='store_true',
                                        help='disable MIOpen find/benchmark mode to pick the best convolution algorithm.')),
    (['--eval-per-calib'], dict(action='store_true', help='evaluate once after every calibration.')),
    (['--exclude-first-conv'], dict(type=str, default="true", choices=['false', 'true'],
                                    help='exclude first convolution layer when quantize. (default: true)')),
    (['--fixed-random-seed'], dict(type=int, default=7,
                                   help='set random_seed for numpy to provide reproducibility. (default: 7)')),
    (['--wino_quantize'], dict(type=str, default="none", choices=['none', 'F23', 'F43', 'F63'],
                               help='quantize weights for Conv2D in Winograd domain (default: none)')),
]
EXTRA_FLAGS = [
    (['--pretrained'], dict(type=str, default="true",
                            help="'true' (look for a local checkpoint, else seeded weights), 'false', or a parameter file")),
    (['--save-qparams'], dict(type=str, default=None,
                              help='write the calibrated parameters (incl. every input_max) to this file')),
    (['--load-qparams'], dict(type=str, default=None,
                              help='load thresholds written by --save-qparams instead of calibrating')),
    (['--no-fuse'], dict(action='store_true',
                         help='keep BatchNorm / ReLU / depthwise convolution as separate library ops '
                              '(default: quantize.fuse.fuse_inference folds them into the fake-quant kernels)')),
    (['--synthetic-on-device'], dict(action='store_true',
                                     help='(synthetic datasets only) generate normalised image batches directly on the GPU '
                                          'instead of image by image on the host, so the reported speed is the '
                                          'network\'s, not the data pipeline\'s; sample values differ from the host '
                                          'pipeline\'s (both are random)')),
    (['--synthetic-resident'], dict(type=int, default=6,
                                    help='(with --synthetic-on-device) distinct synthetic batches kept on the device and cycled '
                                         '(default 6 - a multiple of the default three evaluation lanes, so that a lane meets two '
                                         'of them and replays two graphs in place; 0: draw every batch anew - 19 M normals per '
                                         'ImageNet-shaped batch, a tenth of a MobileNet step)')),
    (['--export-scale-table'], dict(type=str, default=None,
                                    help='after calibration write an ncnn-style int8 scale table (per-channel weight scales '
                                         'after BN folding, one input scale per layer; quantize/freeze/scale_table.py)')),
    (['--eval-streams'], dict(type=int, default=0,
                              help='batches in flight during evaluation, one HIP stream each (quantize/fuse.py keeps the '
                                   'per-forward device state per stream): the ramp and the tail of one batch\'s kernels fill '
                                   'with the next batch\'s work; results are those of one batch at a time.  Calibration passes '
                                   '(update_ema) always run one batch at a time.  (default 0: as bench.py - 4 with '
                                   '--quantize-input-offline, where the layers hand integer codes over, else 3)')),
    (['--eval-graph'], dict(type=int, default=2,
                            help='evaluation of a fused net may replay a hipGraph of the step per lane (static input / label '
                                 'buffers the batches are copied into; the first batch of a lane and a ragged last batch launch '
                                 'eagerly) so that the host no longer launches every kernel of every batch.  2 (default): only '
                                 'where launching from Python would hold the GPU back (host time against device time of the '
                                 'first eager steps); 1: always; 0: never.  Same kernels on the same data either way')),
    (['--strict-global-batch'], dict(action='store_true',
                                     help='(multi-GPU naive calibration) reproduce ONE device that sees the global batch bit '
                                          'for bit: one small all-gather per quantised layer per forward instead of the '
                                          'default single all-reduce per calibration step (dist.py)')),
]


def banner(title, lines=(), show=True):
    if show:
        print('*' * 25 + ' ' + title + ' ' + '*' * 25)
        for line in lines:
            print(line)
        print('*' * (25 * 2 + 2 + len(title)))
        print()


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Simulate for quantization.')
    for flags, kw in REFERENCE_FLAGS + EXTRA_FLAGS:
        parser.add_argument(*flags, **kw)
    opt = parser.parse_args(argv)
    if opt.list_models:
        print("\n".join(get_model_list()))
        raise SystemExit(0)
    if opt.model is None:
        parser.error("--model is required")
    if opt.use_gn:
        parser.error("--use-gn: the model zoo of this build has no GroupNorm variants (gluoncv.nn.GroupNorm is absent)")
    if opt.eval_streams <= 0:
        opt.eval_streams = 4 if opt.quantize_input_offline else 3
    if int(os.environ.get("RANK", "0")) == 0:
        print()
        banner('Settings', ["{0: <25}: {1}".format(k, v) for k, v in vars(opt).items()])
    return opt


# ---- data ------------------------------------------------------------------------------------------------------------------
class UniformSampler(Sampler):
    """`num_per_class` indices of every class, in the order of ONE shuffled pass over the labels (the reference's sampler,
    :151-175: the same `np.random.shuffle` draw, so the same indices for the same seed — and the same on every rank)."""

    def __init__(self, classes, num_per_class, labels):
        self._classes, self._num_per_class = int(classes), int(num_per_class)
        self._labels = np.asarray(labels).astype(np.int64)

    def __len__(self):
        return self._classes * self._num_per_class

    def __iter__(self):
        order = np.arange(len(self._labels))
        np.random.shuffle(order)
        shuffled = self._labels[order]
        keep = []
        for c in range(self._classes):
            where = np.flatnonzero(shuffled == c)[:self._num_per_class]
            if where.size < self._num_per_class:
                raise ValueError("Number of samples for class {} is {} < {}".format(c, float(where.size),
                                                                                    self._num_per_class))
            keep.append(where)
        # first-come order of the shuffled pass (every kept position precedes the point where the last class fills up)
        return iter(order[np.sort(np.concatenate(keep))].tolist())


class DeviceSyntheticLoader(object):
    """Stand-in for DataLoader over the synthetic datasets: N(0,1) "normalised images" and uniform labels generated on
    the device, one generator seed per batch index (so every rank count / batch size sees a well-defined sequence);
    batches are strided across ranks like the real loader's.  `resident` > 0: only that many DISTINCT full batches exist
    (index i shows batch i % resident; generated once, kept on the device) - drawing 19 M normals per batch costs the GPU
    a tenth of a MobileNet step, and the point of this loader is to report the network's speed, not a generator's; the
    ragged last batch is always generated."""

    def __init__(self, n_images, batch_size, shape, classes, ctx, seed, rank=0, world_size=1, resident=0):
        self._n, self._b, self._shape, self._classes = int(n_images), int(batch_size), tuple(shape), int(classes)
        self._dev, self._seed = ctx.torch_device, int(seed)
        self.total_batches = (self._n + self._b - 1) // self._b
        self._mine = range(int(rank), self.total_batches, int(world_size))
        self._resident, self._kept = int(resident), {}
        self.resident_batches = self._resident > 0         # full batches keep their device addresses (evaluate: graphs in place)

    def __len__(self):
        return len(self._mine)

    def _make(self, g, i, b):
        g.manual_seed(self._seed * 1000003 + i)
        X = torch.randn((b,) + self._shape, device=self._dev, generator=g)
        y = torch.randint(0, self._classes, (b,), device=self._dev, generator=g).float()
        return X, y

    def __iter__(self):
        g = torch.Generator(device=self._dev)
        for i in self._mine:
            b = min(self._b, self._n - i * self._b)
            if self._resident > 0 and b == self._b:
                k = i % self._resident
                if k not in self._kept:
                    self._kept[k] = self._make(g, k, b)
                X, y = self._kept[k]
            else:
                X, y = self._make(g, i, b)
            yield mx.nd.NDArray(X), mx.nd.NDArray(y)


def _total_batches(loader):
    """Batches of the WHOLE (unsharded) loader: every rank must take ceil(total / world) calibration steps."""
    if hasattr(loader, "total_batches"):
        return loader.total_batches
    return getattr(loader, "_total_batches", len(loader) * fqdist.world_size())


# ---- evaluation ------------------------------------------------------------------------------------------------------------
class _Lane(object):
    """One evaluation batch in flight: a HIP stream, and - once the lane has run a forward eagerly - hipGraphs of the step
    (forward + the evaluation counters), replayed for every later full batch.  Launching the ~30-100 kernels of a fused step
    from Python takes the host as long as the GPU needs to run them (bench.py measures it); a replay is one call.  Where the
    graph reads its batch:
      * a loader whose batches live on the device at stable addresses (`loader.resident_batches`: the synthetic on-device
        loader) is read IN PLACE - one graph per (lane, batch tensor), at most `MAX_GRAPHS` of them;
      * any other batch is copied into the lane's static input buffer - for a host batch that IS its host-to-device copy,
        for a device batch one extra device copy (77 MB for 128 ImageNet images: ~6 % of a MobileNet step)."""
    MAX_GRAPHS = 16
    capture_s, captures = 0.0, 0        # host time spent capturing + instantiating, all lanes (FQ_EVAL_TIMING=1 prints it)
    replay_s = 0.0                      # host time inside hipGraphLaunch, all lanes

    def __init__(self, dev, stream):
        self.dev, self.stream = dev, stream
        self.graphs, self.full_shape, self.failed = {}, None, False
        self.capture_stream = None
        self.pool = None
        self.eager_done = 0

    def _capture(self, xbuf, ybuf, step):
        """The step captured with the graph's own begin / end calls on the lane's stream - NOT the `torch.cuda.graph`
        context, whose device-wide synchronise + gc.collect + empty_cache cost a 0.4 s evaluation pass a third of its time."""
        t0 = time.perf_counter()
        try:
            return self._capture_impl(xbuf, ybuf, step)
        finally:
            _Lane.capture_s += time.perf_counter() - t0
            _Lane.captures += 1

    def _capture_impl(self, xbuf, ybuf, step):
        g = torch.cuda.CUDAGraph()
        side = self.stream
        if side is None:                             # one lane on the default stream: capture needs a stream of its own
            side = self.capture_stream = self.capture_stream or torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
        try:
            with torch.cuda.stream(side):
                # one memory pool per lane: its graphs never overlap in time, and consecutive batches of a lane then run over the
                # SAME intermediate buffers (what bench.py does) instead of one private set of activations per captured graph
                if self.pool is None:
                    self.pool = torch.cuda.graph_pool_handle()
                g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
                try:
                    step(xbuf, ybuf)
                finally:
                    g.capture_end()
            if self.stream is None:
                torch.cuda.current_stream(self.dev).wait_stream(side)
        except Exception as e:                       # capture is an optimisation: stay with eager launches, and say so once
            self.failed = True
            torch.cuda.synchronize(self.dev)
            print("[eval] hipGraph capture failed (%s: %s): evaluation launches eagerly" % (type(e).__name__, str(e)[:120]))
            return None
        return g

    def replay(self, x, y, step, resident):
        """True when the batch (x: fp32 tensor on the host or the device, y: labels of any dtype) went through a graph."""
        if self.failed or self.eager_done == 0:
            return False
        if self.full_shape is None:
            self.full_shape = tuple(x.shape)
        if tuple(x.shape) != self.full_shape:        # the ragged last batch: not worth a graph of its own
            return False
        in_place = resident and x.is_cuda
        key = (x.data_ptr(), y.data_ptr()) if in_place else "static"
        ent = self.graphs.get(key)
        if ent is None:
            if len(self.graphs) >= self.MAX_GRAPHS:
                return False
            xbuf = x if in_place else torch.empty(x.shape, dtype=torch.float32, device=self.dev)
            ybuf = torch.empty(y.shape, dtype=torch.long, device=self.dev)
            if not in_place:
                xbuf.copy_(x, non_blocking=True)
            ybuf.copy_(y, non_blocking=True)
            g = self._capture(xbuf, ybuf, step)
            if g is None:
                return False
            ent = self.graphs[key] = (g, xbuf, ybuf)
        g, xbuf, ybuf = ent
        if not in_place:
            xbuf.copy_(x, non_blocking=True)
            ybuf.copy_(y, non_blocking=True)         # (in place: the labels of a resident batch never change)
        t0 = time.perf_counter()
        g.replay()
        _Lane.replay_s += time.perf_counter() - t0
        return True


class _EvalState(object):
    """What an evaluation pass builds once and later passes over the same net in the same mode take over: the lanes (streams,
    captured graphs, static buffers), the device counters every graph writes to, the decision whether to replay at all.
    Kept on the net, valid while `convert.mode_epoch()` stands (quantize_input / enable / fix_params / fusing change which
    kernels a forward launches: the graphs of an older epoch are dropped; calibration only changes values the kernels read -
    but an evaluation state is also dropped after a calibration step, see `_eval_state`).  `quantize_input` / `enable` / `disable`
    move the epoch only when they change a flag, so the CLI's repeated calls before every epoch keep the graphs."""

    def __init__(self, dev, n_lanes, num_class, epoch):
        self.epoch = epoch
        self.counters = torch.zeros(2 + 2 * num_class, dtype=torch.float32, device=dev)
        self.lanes = [_Lane(dev, torch.cuda.Stream(dev) if n_lanes > 1 else None) for _ in range(n_lanes)] \
            if dev.type == "cuda" else None
        self.use_graph = None              # undecided / True / False ("auto": decided from the first eager batches)
        self.passes = 0


def _epochs():
    from quantization.mxnet_amd.quantize.convert.convert import mode_epoch
    from quantization.mxnet_amd.mx.gluon.parameter import write_epoch
    return (mode_epoch(), write_epoch())


def _eval_state(net, dev, n_lanes, num_class, update_ema):
    cache = net.__dict__.setdefault("_fq_eval_states", {})
    key = (str(dev), n_lanes, num_class, bool(update_ema))
    st = cache.get(key)
    # (a calibration pass of a fake-BN net writes running statistics with set_data in every step: its graphs hold those
    # tensors' addresses and the writes are in place, but the epoch moves - such a net simply captures once per pass)
    # ... and an EVALUATION state also dies with the library's own in-place threshold updates (ops.state_epoch(): update_ema
    # writes through raw pointers): captured graphs bake in host decisions made from threshold VALUES (convert_conv2d.
    # _same_quantiser), which a calibration between two evaluate() calls may have changed
    epoch = _epochs() + (() if update_ema else (ops.state_epoch(),))
    if st is None or st.epoch != epoch:
        st = cache[key] = _EvalState(dev, n_lanes, num_class, epoch)
    return st


# `--eval-graph 2` (auto, the default): replay only where launching a step from Python cannot keep up with the GPU.  The host
# time of a lane's first (eager) step is compared with the device time of the same step (an event pair); with S lanes the host
# must issue S steps while the device runs one lane's step, so eager launches suffice when host < margin x device.  mobilenet1.0
# (45 launches per step) stays eager - 109.6 k images/s against 105.9 k with its captures paid (profiles/r4_cli_vs_bench.txt);
# MobileNetV2 (~100 launches) replays: 121 k against 54 k.
_AUTO_GRAPH_MARGIN = float(os.environ.get("FQ_EVAL_AUTO_MARGIN", "0.8"))


def evaluate(net, num_class, dataloader, ctx, update_ema=False, tqdm_desc="Eval", streams=1, graph=False):
    """One pass over `dataloader`: top-1 accuracy and class-averaged accuracy (the quantities of the reference's
    `evaluate`, :122-148).  The counters [n_correct, total, correct[c], label[c]] live on the device (fq_eval_counters; in
    the classifier's own launch when the net is fused: quantize.fuse.EvalHead) and cross the ranks in one all-reduce; with
    `update_ema` every batch is a calibration step.  `streams` > 1 (evaluation only): that many batches in flight.  `graph`
    (evaluation of a fused net only; 1 = always, 2 = where the host would otherwise be the bottleneck, see
    `_AUTO_GRAPH_MARGIN`): each lane replays a hipGraph of its step over static buffers the batches are copied into (the
    first batch of a lane and batches of another shape - the ragged last one - launch eagerly); same kernels on the same
    data, so the same counters.  Lanes, graphs and counters are kept on the net and reused by later passes in the same mode
    (`_EvalState`): calibration epochs and `--eval-per-calib` evaluations capture once."""
    from quantization.mxnet_amd.quantize import fuse as _fuse
    dev = ctx.torch_device
    on_gpu = dev.type == "cuda"
    n_lanes = streams if streams > 1 and not update_ema and on_gpu else 1
    if n_lanes > 1 and _fuse.library_gemm_blocks(net):
        n_lanes = 1                                        # (library GEMMs in the net: one batch at a time, see fuse.library_gemm_blocks)
    state = _eval_state(net, dev, n_lanes, num_class, update_ema)
    counters = state.counters
    counters.zero_()
    seen, started = 0, time.perf_counter()
    steps = fqdist.calibration_steps(_total_batches(dataloader)) if update_ema and fqdist.group_is_live() else None
    done = 0
    lanes = state.lanes
    # (a calibration pass replays too when no process group exists: its step - forward with online scales + update_ema -
    # holds no collective and no host synchronisation; with a group the step's all-reduce must stay outside any capture)
    graph_ok = on_gpu and hasattr(net, "_fq_arena_hooks") and not (update_ema and fqdist.group_is_live())
    mode = int(graph) if graph_ok else 0
    if mode == 1:
        state.use_graph = True
    elif mode == 0:
        state.use_graph = False
    head = _fuse.eval_head(net, counters) if on_gpu else None
    replayed = 0
    resident = bool(getattr(dataloader, "resident_batches", False))
    timing = {"eager_host_s": [], "eager_dev_ev": [], "first_s": 0.0}
    captures0, capture_s0, replay_s0 = _Lane.captures, _Lane.capture_s, _Lane.replay_s
    steady = {"t": None, "seen": 0, "captures": _Lane.captures, "capture_at": 0}   # from where the steady-state figure counts

    def step(xt, labels):
        """forward + counters of one batch on the current stream"""
        if head is not None:
            head.labels = labels
        logits = net(mx.nd.NDArray(xt))
        if update_ema:
            net.update_ema()
        if head is None or not head.take():
            ops.eval_counters(logits._t, labels, counters)

    producer = torch.cuda.current_stream(dev) if on_gpu else None      # the stream the loader's device work is issued on

    def decide():
        """auto mode: host time against device time of the eager steps seen so far (the first one of a pass that freezes the
        weights is left out when there are others)."""
        if state.use_graph is not None or not timing["eager_dev_ev"]:
            return
        torch.cuda.synchronize(dev)
        dev_ms = [a.elapsed_time(b) for a, b in timing["eager_dev_ev"]]
        host_ms = [h * 1e3 for h in timing["eager_host_s"]]
        if len(dev_ms) > 1:
            dev_ms, host_ms = dev_ms[1:], host_ms[1:]
        d, h = min(dev_ms), min(host_ms)
        # the device finishes a step every ~0.9 d with the lanes overlapping; the one host thread needs h per step
        state.use_graph = not (h < _AUTO_GRAPH_MARGIN * 0.9 * d)
        state.decision = "host %.2f ms / device %.2f ms per eager step, %d lane(s) -> %s" % (
            h, d, n_lanes, "replay from hipGraphs" if state.use_graph else "eager launches keep up")

    def run_batch(lane, X, y, index):
        """One batch on its lane (called with the lane's stream current).  Returns True when it went through a graph."""
        side = lane.stream if lane is not None else None
        if side is not None:
            # the lanes are non-blocking streams: what the producer stream has issued so far - the calibrated thresholds, this
            # batch if the loader made it on the device - must be complete before the lane reads it, and the batch's memory
            # must not go back to the producer's pool while the lane still reads it
            side.wait_stream(producer)
            for t in (X._t, y._t):
                if t.is_cuda:
                    t.record_stream(side)
        if state.use_graph and lane.eager_done > 0 and lane.replay(X._t, y._t, step, resident and not update_ema):
            return True
        measure = on_gpu and state.use_graph is None and len(timing["eager_dev_ev"]) < max(2, n_lanes)
        if measure:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            t0 = time.perf_counter()
        step(X.as_in_context(ctx)._t, y.as_in_context(ctx)._t.long())
        if measure:
            timing["eager_host_s"].append(time.perf_counter() - t0)
            ev[1].record()
            timing["eager_dev_ev"].append(ev)
        if lane is not None:
            lane.eager_done += 1
        return False

    try:
        with tqdm(total=len(dataloader), desc=tqdm_desc, disable=fqdist.rank() != 0) as bar:
            for X, y in dataloader:
                # ONE host thread feeds every lane: a replay costs it 0.04 ms against the ~1 ms the GPU needs per batch (a
                # thread per lane was measured: 126.2 k vs 127.3 k images/s on MobileNetV2, tools/cli_lane_probe.py)
                lane = lanes[done % n_lanes] if lanes else None
                side = lane.stream if lane is not None else None
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()), \
                        (ops.batches_in_flight() if side is not None else contextlib.nullcontext()):
                    replayed += bool(run_batch(lane, X, y, done))
                seen += int(y._t.numel())
                done += 1
                bar.update(1)
                if on_gpu and done == 1 and state.passes == 0:
                    # what the first forward creates once - frozen weights, weight codes, folded BatchNorm constants - is read
                    # by the forwards on the other streams: wait for it
                    torch.cuda.synchronize(dev)
                    timing["first_s"] = time.perf_counter() - started
                if on_gpu and mode == 2 and state.use_graph is None and done == max(2, n_lanes):
                    decide()
                if _Lane.captures != steady["captures"]:
                    steady["captures"], steady["capture_at"] = _Lane.captures, done
                if on_gpu and steady["t"] is None and done >= 2 * n_lanes and done - steady["capture_at"] >= n_lanes \
                        and (mode != 2 or state.use_graph is not None):
                    # set-up is behind us (first forwards, the decision, the captures so far): the steady state starts
                    torch.cuda.synchronize(dev)
                    steady["t"], steady["seen"] = time.perf_counter(), seen
    finally:
        if head is not None:
            head.release()
    while steps is not None and done < steps:          # this rank's shard ran out first: keep the collectives in step
        fqdist.empty_calibration_step(net)
        done += 1
    if on_gpu:
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - started
    state.passes += 1
    # what this pass itself wrote - the freezing forward stores the quantised weights with set_data (convert_conv2d.py:101-108) -
    # happened before its first capture (the first batch is launched eagerly and waited for): the graphs belong to the epoch the
    # pass ENDS in
    state.epoch = _epochs() + (() if update_ema else (ops.state_epoch(),))
    fqdist.allreduce_eval_counters(counters)
    c = counters.cpu().numpy()
    evaluate.last_images_per_sec = seen * fqdist.world_size() / max(elapsed, 1e-9)
    evaluate.last_steady_images_per_sec = None
    if steady["t"] is not None and seen > steady["seen"]:
        evaluate.last_steady_images_per_sec = (seen - steady["seen"]) * fqdist.world_size() / max(
            started + elapsed - steady["t"], 1e-9)
    evaluate.last_replayed = replayed
    evaluate.last_setup = {"first_batch_s": timing["first_s"], "captures": _Lane.captures - captures0,
                           "capture_s": _Lane.capture_s - capture_s0, "graph_mode": mode, "use_graph": state.use_graph,
                           "decision": getattr(state, "decision", None), "pass_of_this_state": state.passes}
    if os.environ.get("FQ_EVAL_TIMING", "0") == "1" and fqdist.rank() == 0:
        print("[eval] %s: %d images in %.3f s; set-up: first batch %.0f ms, %d capture(s) %.0f ms; %s; %d of %d batches replayed, "
              "%.0f us of host time per hipGraphLaunch"
              % (tqdm_desc, seen, elapsed, timing["first_s"] * 1e3, _Lane.captures - captures0,
                 (_Lane.capture_s - capture_s0) * 1e3, getattr(state, "decision", None) or "graph mode %d" % mode, replayed, done,
                 (_Lane.replay_s - replay_s0) * 1e6 / max(replayed, 1)))
    per_class = c[2:2 + num_class] / (c[2 + num_class:] + 1e-10)
    return float(c[0] / max(c[1], 1.0)), float(per_class.mean())


evaluate.last_images_per_sec = 0.0
evaluate.last_steady_images_per_sec = None
evaluate.last_replayed = 0
evaluate.last_setup = {}


# ---- the program -----------------------------------------------------------------------------------------------------------
class Simulation(object):
    def __init__(self, opt, ctx, rank=0, world=1):
        self.opt, self.ctx, self.rank, self.world = opt, ctx, rank, world
        # the calibration collectives are attached whenever a process group exists - with several ranks, or with ONE rank of
        # a forced group (FQ_DIST_FORCE_GROUP=1: how the RCCL branch is exercised end to end on a one-GPU box)
        self.collective = world > 1 or fqdist.group_is_live()
        self.chief = rank == 0
        self.classes = 10 if opt.dataset == 'cifar10' else 1000
        self.signed = opt.input_signed == 'true'
        self.net = None
        self.last_result = None
        self.data_source = 'unknown'
        self.weights_source = 'unknown'

    # -- model ---------------------------------------------------------------------------------------------------------
    def build_net(self):
        opt = self.opt
        pretrained = {"true": True, "false": False}.get(opt.pretrained.lower(), opt.pretrained)
        zoo_args = dict(pretrained=pretrained, classes=self.classes)
        if opt.model.startswith('vgg'):
            zoo_args['batch_norm'] = opt.batch_norm
        if opt.model.startswith('resnext'):
            zoo_args['use_se'] = opt.use_se
        if opt.last_gamma:
            zoo_args['last_gamma'] = True
        self.net = get_model(opt.model, **zoo_args)
        from quantization.mxnet_amd.mx.gluon.model_zoo import find_checkpoint
        ckpt = pretrained if isinstance(pretrained, str) else (find_checkpoint(opt.model.lower()) if pretrained else None)
        self.weights_source = ckpt if ckpt else 'seeded He-normal (no checkpoint found)'
        if opt.print_model:
            banner(opt.model, [repr(self.net)], self.chief)

    def excluded_blocks(self):
        net, name, out = self.net, self.opt.model, []
        if self.opt.exclude_first_conv == 'true':
            out += [net.features[0], net.features[1]]
        if name.startswith('mobilenetv2_'):
            out.append(net.output[0])
        if name.startswith('cifar_resnet'):
            first_unit = net.features[2][0].body
            out += [first_unit[0], first_unit[1]]
        return out

    def quantise_net(self):
        opt = self.opt
        shared = dict(quantize_input=True, input_signed=self.signed, input_width=opt.input_bits_width,
                      weight_width=opt.weight_bits_width, quant_type=opt.quant_type)
        converters = {
            nn.Conv2D: convert.gen_conv2d_converter(fake_bn=opt.merge_bn, wino_quantize=opt.wino_quantize, **shared),
            nn.Dense: convert.gen_dense_converter(**shared),
            nn.BatchNorm: convert.bypass_bn if opt.merge_bn else None,
            nn.Activation: None,
        }
        skip = self.excluded_blocks()
        banner('Exclude blocks', [b.name for b in skip], self.chief)
        convert.convert_model(self.net, exclude=skip, convert_fn=converters)
        qparams_init(self.net)
        self.net.collect_params().reset_ctx(self.ctx)
        if not opt.no_fuse and self.ctx.device_type == "gpu":
            from quantization.mxnet_amd.quantize import fuse
            n_fused = fuse.fuse_inference(self.net)
            if self.chief:
                print("[fuse] %d BatchNorm / depthwise blocks folded into fused HIP passes (--no-fuse to disable)" % n_fused)

    # -- data ----------------------------------------------------------------------------------------------------------
    def _transform(self):
        T = vision.transforms
        if self.opt.dataset == 'imagenet':
            return T.Compose([T.Resize(256, keep_ratio=True), T.CenterCrop(224), T.ToTensor(),
                              T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
        return T.Compose([T.ToTensor(), T.Normalize([0.4914, 0.4822, 0.4465], [0.2023, 0.1994, 0.2010])])

    def make_loaders(self):
        opt = self.opt
        shard = fqdist.shard_loader_kwargs()
        dataset = vision.ImageNet if opt.dataset == 'imagenet' else vision.CIFAR10
        needs_calibration_data = opt.quantize_input_offline and not opt.load_qparams
        eval_raw = dataset(train=False)
        eval_set = eval_raw.transform_first(self._transform())
        self.data_source = eval_raw.source
        self.train_loader = None
        if opt.synthetic_on_device:
            if eval_raw.source != 'synthetic':
                raise SystemExit("--synthetic-on-device asks for generated batches but %s holds the dataset: drop the flag (or "
                                 "point FQ_IMAGENET_ROOT / FQ_CIFAR10_ROOT elsewhere)" % eval_raw.source[5:])
            self.data_source = 'synthetic (generated on the device)'
            side = 224 if opt.dataset == 'imagenet' else 32
            self.eval_loader = DeviceSyntheticLoader(len(eval_set), opt.batch_size, (3, side, side), self.classes,
                                                     self.ctx, 7, resident=opt.synthetic_resident, **shard)
            if needs_calibration_data:
                self.train_loader = DeviceSyntheticLoader(self.classes * opt.num_sample, opt.batch_size,
                                                          (3, side, side), self.classes, self.ctx, 11, **shard)
            return
        self.eval_loader = DataLoader(dataset=eval_set, batch_size=opt.batch_size, num_workers=opt.num_workers,
                                      last_batch='keep', **shard)
        if needs_calibration_data:
            train_set = dataset(train=True).transform_first(self._transform())
            labels = [item[1] for item in train_set._data.items] if opt.dataset == 'imagenet' \
                else train_set._data._label
            sampler = UniformSampler(self.classes, opt.num_sample, labels)
            self.train_loader = DataLoader(dataset=train_set, batch_size=opt.batch_size, sampler=sampler,
                                           num_workers=opt.num_workers, last_batch='keep', **shard)
            self.train_loader._total_batches = (len(sampler) + opt.batch_size - 1) // opt.batch_size

    # -- calibration ---------------------------------------------------------------------------------------------------
    def calibrate_kl(self):
        """Histograms of every quantised block's fp32 input, then the KL threshold search (reference :296-315)."""
        net, opt = self.net, self.opt
        title = ' KL Calibration '
        if self.chief:
            print('*' * 25 + title + '*' * 25)
        net.disable_quantize()                   # fp32 inputs and fp32 weights while collecting
        levels = 2 ** (opt.input_bits_width - 1 if self.signed else opt.input_bits_width)
        bins = 2048
        hists, ranges = collect_feature_maps(net, bins=bins, loader=self.train_loader, ctx=self.ctx,
                                             sync=fqdist.kl_sync if self.collective else None)
        blocks = net.collect_quantized_blocks()
        best = kl_calibrate_many([hists[b] for b in blocks], levels=levels, min_bins=levels, bins=bins,
                                 device=self.ctx.torch_device)
        for i, (blk, best_bins) in enumerate(zip(blocks, best)):
            threshold = (best_bins + 0.5) * (ranges[blk] / bins)
            if self.chief:
                print(f"({i+1}/{len(blocks)})\tBest threshold for {blk.name}: {threshold}")
            blk.input_max.set_data(nd.array([threshold], ctx=self.ctx))
        net.enable_quantize()
        if self.chief:
            print('*' * (25 * 2 + len(title)))
            print()

    def calibrate_naive(self):
        """EMA of the online statistic over `--calib-epoch` passes of the calibration sample (reference :317-334)."""
        net, opt = self.net, self.opt
        title = ' Naive Calibration '
        if self.chief:
            print('*' * 25 + title + '*' * 25)
        if self.collective:
            fqdist.attach_calibration_sync(net, opt.batch_size, strict=opt.strict_global_batch)
        for epoch in range(1, opt.calib_epoch + 1):
            net.quantize_input(enable=True, online=True)      # integer inputs and weights, ranges from the current batch
            evaluate(net, self.classes, self.train_loader, ctx=self.ctx, update_ema=True,
                     tqdm_desc="Calib[{}/{}]".format(epoch, opt.calib_epoch), graph=opt.eval_graph)
            if opt.eval_per_calib:
                if self.collective:
                    fqdist.detach_calibration_sync(net)       # offline evaluation exchanges nothing per batch
                net.quantize_input(enable=True, online=False)
                self.report(*evaluate(net, self.classes, self.eval_loader, ctx=self.ctx,
                                      tqdm_desc="Eval[{}/{}]".format(epoch, opt.calib_epoch), streams=opt.eval_streams,
                                      graph=opt.eval_graph))
                if self.chief:
                    print()
                if self.collective and epoch < opt.calib_epoch:
                    fqdist.attach_calibration_sync(net, opt.batch_size, strict=opt.strict_global_batch)
        if self.collective:
            fqdist.detach_calibration_sync(net)
        if self.chief:
            for blk in net.collect_quantized_blocks():
                print(f"Best threshold for {blk.name}: {blk.input_max.data().asscalar()}")
            print('*' * (25 * 2 + len(title)))
            print()

    # -- evaluation ----------------------------------------------------------------------------------------------------
    def report(self, acc, avg_acc):
        self.last_result = (acc, avg_acc)
        if self.chief:
            print('{0: <8}: {1:2.2f}%'.format('acc', acc * 100))
            print('{0: <8}: {1:2.2f}%'.format('avg_acc', avg_acc * 100))
            # the figure comparable with bench.py's (which warms up before it times): batches per second once the pass is set up -
            # the first forward of a process loads kernels and freezes the weights (~0.1 s), a replayed evaluation captures its
            # graphs; the whole-pass figure, set-up included, follows
            steady, su = evaluate.last_steady_images_per_sec, evaluate.last_setup
            if steady:
                print('{0: <8}: {1:.1f} images/sec on {2} GPU(s) once set up (first forward {3:.0f} ms, {4} graph capture(s) {5:.0f} '
                      'ms; {6})'.format('speed', steady, self.world, su.get("first_batch_s", 0.0) * 1e3, su.get("captures", 0),
                                        su.get("capture_s", 0.0) * 1e3, su.get("decision") or "--eval-graph %d" % su.get("graph_mode", 0)))
                print('{0: <8}: {1:.1f} images/sec over the whole pass, set-up included'.format('', evaluate.last_images_per_sec))
            else:
                print('{0: <8}: {1:.1f} images/sec on {2} GPU(s)'.format('speed', evaluate.last_images_per_sec, self.world))
            # where the images and the weights came from: an accuracy on synthetic images / random weights means nothing
            print('{0: <8}: {1}'.format('data', self.data_source))
            print('{0: <8}: {1}'.format('weights', self.weights_source))

    def final_evaluation(self, online):
        self.net.fix_params()
        self.net.quantize_input(enable=True, online=online)
        acc, avg_acc = evaluate(self.net, self.classes, self.eval_loader, ctx=self.ctx, streams=self.opt.eval_streams,
                                graph=self.opt.eval_graph)
        if self.chief:
            print('*' * 25 + ' Result ' + '*' * 25)
        self.report(acc, avg_acc)
        if self.chief:
            print('*' * (25 * 2 + len(' Result ')))
            print()
        return acc, avg_acc

    def execute(self):
        opt = self.opt
        np.random.seed(opt.fixed_random_seed)        # identical on every rank: same weights, same sampler draws
        torch.backends.cudnn.benchmark = False       # MIOpen find mode: measured no gain, +60 s start-up
        self.build_net()
        self.quantise_net()
        self.make_loaders()
        if not opt.quantize_input_offline:
            return self.final_evaluation(online=True) + (self.net,)
        if opt.load_qparams:
            self.net.load_parameters(opt.load_qparams, ctx=self.ctx, allow_missing=True, ignore_extra=True)
        elif opt.calib_mode == "kl":
            self.calibrate_kl()
        else:
            self.calibrate_naive()
        if opt.save_qparams and self.chief:
            self.net.save_parameters(opt.save_qparams)
        if opt.export_scale_table and self.chief:
            from quantization.mxnet_amd.quantize.freeze import export_scale_table
            export_scale_table(self.net, opt.export_scale_table, weight_width=8, input_width=8,
                               json_path=opt.export_scale_table + ".json")
        if opt.eval_per_calib and self.last_result is not None:
            return self.last_result + (self.net,)      # already evaluated after the last calibration pass
        return self.final_evaluation(online=False) + (self.net,)


def run(opt, ctx, rank=0, world=1):
    """-> (acc, avg_acc, net)"""
    return Simulation(opt, ctx, rank, world).execute()


def main(argv=None):
    rank, local_rank, world = fqdist.init()
    opt = parse_args(argv)
    ctx = gpu(local_rank) if world > 1 else (gpu(opt.use_gpu) if opt.use_gpu != -1 else cpu())
    if ctx.device_type == "cpu":
        raise SystemExit("error: this build runs the fake-quant path on an MI355X only (no CPU fallback); "
                         "pass --use-gpu=<id>")
    try:
        return run(opt, ctx, rank, world)
    finally:
        fqdist.shutdown()


if __name__ == "__main__":
    main()
