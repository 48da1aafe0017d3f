"""One process per GPU: batch sharding and the (tiny) collectives of calibration and evaluation.

The reference is single-process, single-device (no kvstore / NCCL / multi-ctx split anywhere), so
everything here is additive.  The path shards over independent images; the exchanges are:

  evaluation        one all-reduce(sum) of the accuracy counters at the end (simulate_quantization.py:123-147)
  online eval       none: each rank's local batch is "the batch" (== the reference run with --batch-size=local)
  naive-EMA calib   DEFAULT: each rank runs its forward on its local batch (online scales from the local statistic, as in
                    online evaluation); at the step's `update_ema` ONE all-reduce(sum) of L+1 doubles — per layer the
                    fp64 sum of its per-sample maxima, plus the local sample count — gives every rank
                    current_input_max[l] = fp32(sum)/fp32(n) = the batch mean of the GLOBAL batch (the `.mean()` of
                    convert_conv2d.py:56) and the identical EMA update (convert.py:66-70): replicas stay bit-identical.
                    Ranks without a batch in a step (ragged batch counts) contribute an empty record, so every rank
                    issues the same number of collectives.
                    STRICT (--strict-global-batch): additionally reproduce what ONE device would compute on the global
                    batch bit for bit — there a layer's online scale depends on its batch-mates, so the global statistic
                    is needed between a layer's statistic pass and its apply pass: one all-gather of (1 + n_local)
                    floats per quantised layer per forward (27-53 latency-bound collectives).
  KL calib          broadcast of the first global batch's ranges `fm_max[L]` from rank 0 (the rank that holds global
                    batch 0: distribution_calibrate.py:97-101 fixes the range with the FIRST batch), and ONE
                    all-reduce(sum) of the exact int64 histograms [L x 2048] (434 KB for ResNet-50) at the end
                    (:103-104): bit-identical to one device walking the same batches.

Transport: `torch.distributed` — backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.  All messages
are latency-bound (<= 434 KB), so they go on the compute stream with RCCL's defaults; no bucketing is needed.

Second transport, FQ_DIST_BACKEND=fqcomm: the library's own RCCL communicator behind the C ABI (include/fakequant.h:
fq_comm_unique_id / fq_comm_init / fq_allreduce_* / fq_comm_destroy) - what a host WITHOUT torch.distributed binds, i.e. the
reference's MXNet process with the ctypes stub of INTEGRATION.md.  Rank and world size come from the same environment
(RANK / WORLD_SIZE), rank 0's 128-byte unique id travels through a file (FQ_COMM_ID_FILE, default
$TMPDIR/fq_comm_id_<MASTER_PORT>).  Every exchange of this module is expressed through its ONE primitive, the in-place
all-reduce: a broadcast is the sum with zeros on the other ranks, an all-gather the sum of zero-padded slices (exact: x + 0).
Same collectives per step, same bit-identical replicas; exercised with one rank on the one-GPU boxes of this pool
(tests/test_gpu_rccl.py), like the torch.distributed "nccl" path.

Test knobs (the pool's GPU boxes have ONE device and RCCL refuses two ranks on it): FQ_DIST_BACKEND=gloo selects gloo
although a GPU is present — device tensors are then staged through the host around each collective — and
FQ_DIST_SHARE_GPU=1 puts every rank on device 0, so that the N > 1 flows run through the real kernels on such a box.
"""
import os

import torch
import torch.distributed as dist

from . import ops

__all__ = ["init", "is_distributed", "group_is_live", "rank", "world_size", "shard_loader_kwargs", "attach_calibration_sync", "detach_calibration_sync", "empty_calibration_step", "calibration_steps",
           "kl_sync", "allreduce_eval_counters", "shutdown"]


_FQ = {"on": False, "rank": 0, "world": 1}          # the fqcomm transport's state (the library holds the communicator)


def _td_live():
    return dist.is_available() and dist.is_initialized()


def is_distributed():
    return world_size() > 1


def group_is_live():
    """A process group exists (also with ONE rank: `FQ_DIST_FORCE_GROUP=1`, or a caller's own init_process_group) - the
    collectives of calibration and evaluation are then really issued, which is how the RCCL branch is exercised end to end
    on a one-GPU box (tests/test_gpu_rccl.py)."""
    return _FQ["on"] or _td_live()


def rank():
    if _FQ["on"]:
        return _FQ["rank"]
    return dist.get_rank() if _td_live() else 0


def world_size():
    if _FQ["on"]:
        return _FQ["world"]
    return dist.get_world_size() if _td_live() else 1


def _fqcomm_init(rank_, world, local):
    import tempfile
    import time
    torch.cuda.set_device(local)
    path = os.environ.get("FQ_COMM_ID_FILE") or os.path.join(
        tempfile.gettempdir(), "fq_comm_id_%s" % os.environ.get("MASTER_PORT", "29533"))
    if rank_ == 0:
        uid = ops.comm_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
    else:
        deadline = time.time() + float(os.environ.get("FQ_COMM_ID_TIMEOUT_S", "120"))
        while not os.path.exists(path):
            if time.time() > deadline:
                raise RuntimeError("fqcomm: rank 0's unique id did not appear at %s" % path)
            time.sleep(0.05)
        with open(path, "rb") as f:
            uid = f.read()
    ops.comm_init(rank_, world, uid)
    _FQ.update(on=True, rank=rank_, world=world, id_file=path if rank_ == 0 else None)
    # (the file may only go once every rank has read it: the first collective says so)
    all_reduce(torch.zeros(1, dtype=torch.float32, device=torch.device("cuda", local)))
    torch.cuda.synchronize()
    if rank_ == 0:
        try:
            os.remove(path)
        except OSError:
            pass


def init(backend=None):
    """Join the job described by torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world).  A no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("FQ_DIST_FORCE_GROUP", "0") != "1":
        return 0, int(os.environ.get("LOCAL_RANK", "0")), 1
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FQ_DIST_SHARE_GPU", "0") == "1":
        local = 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world <= 1:                               # a one-rank group (FQ_DIST_FORCE_GROUP=1): env:// needs the full set
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if backend is None:
        backend = os.environ.get("FQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "fqcomm":
        _fqcomm_init(int(os.environ.get("RANK", "0")), max(world, 1), local)
        return _FQ["rank"], local, _FQ["world"]
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return dist.get_rank(), local, dist.get_world_size()


def shutdown():
    if _FQ["on"]:
        all_reduce(torch.zeros(1, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device())))
        torch.cuda.synchronize()
        ops.comm_destroy()
        _FQ.update(on=False, rank=0, world=1)
        return
    if _td_live():
        dist.barrier()
        dist.destroy_process_group()


# ---- transport ---------------------------------------------------------------------------------------------------------
def _via_host(t):
    _not_while_capturing(t)
    return t.is_cuda and not _FQ["on"] and dist.get_backend() == "gloo"


def _not_while_capturing(t):
    """Collectives are issued on the compute stream, next to forwards that an evaluation loop may be capturing into
    hipGraphs (bench.py, the CLI's `evaluate(graph=True)`): a collective that landed inside a capture would be baked into the
    graph and replayed with every batch - or fail inside RCCL with an error that names neither.  Calibration never captures,
    evaluation exchanges its counters after the last replay; anything else is refused here."""
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("a torch.distributed collective was issued while the current HIP stream is capturing a graph: "
                           "calibration steps and the final counter all-reduce must run outside hipGraph capture")


_COLLECTIVES = {}       # kind -> [calls, payload bytes]: what this module issued (bench.py reports it per step)


def _note(kind, t):
    rec = _COLLECTIVES.setdefault(kind, [0, 0])
    rec[0] += 1
    rec[1] += t.numel() * t.element_size()


def collective_stats(reset=False):
    """{kind: {"calls", "bytes"}} of the collectives issued through this module since the last reset - the data-path exchanges
    of a calibration step / a KL collection / the final counter sum, for checking a multi-GPU line in one pass."""
    out = {k: {"calls": v[0], "bytes": v[1]} for k, v in _COLLECTIVES.items()}
    if reset:
        _COLLECTIVES.clear()
    return out


def _fq_all_reduce(t, op):
    _not_while_capturing(t)
    code = ops.COMM_MAX if op == dist.ReduceOp.MAX else ops.COMM_SUM
    if op not in (dist.ReduceOp.SUM, dist.ReduceOp.MAX):
        raise ValueError("the fqcomm transport reduces with SUM or MAX (got %r)" % (op,))
    if t.dtype in (torch.float32, torch.float64, torch.int64) and t.is_contiguous():
        ops.comm_allreduce(t, code)
    else:                                        # (int32 counters and the like: through an exact wider copy)
        w = t.contiguous().to(torch.int64 if not t.dtype.is_floating_point else torch.float64)
        ops.comm_allreduce(w, code)
        t.copy_(w.to(t.dtype))
    return t


def all_reduce(t, op=None):
    op = dist.ReduceOp.SUM if op is None else op
    _note("all_reduce", t)
    if _FQ["on"]:
        return _fq_all_reduce(t, op)
    if _via_host(t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def broadcast(t, src=0):
    _note("broadcast", t)
    if _FQ["on"]:
        if _FQ["rank"] != src:
            t.zero_()
        return _fq_all_reduce(t, dist.ReduceOp.SUM)
    if _via_host(t):
        h = t.detach().cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)
    return t


def all_gather_into(gathered, piece):
    _note("all_gather", gathered)
    if _FQ["on"]:
        n = piece.numel()
        gathered.zero_()
        gathered.reshape(-1)[_FQ["rank"] * n:(_FQ["rank"] + 1) * n].copy_(piece.reshape(-1))
        return _fq_all_reduce(gathered, dist.ReduceOp.SUM)
    if _via_host(piece):
        h = torch.empty(gathered.shape, dtype=gathered.dtype)
        dist.all_gather_into_tensor(h, piece.detach().cpu())
        gathered.copy_(h)
    else:
        dist.all_gather_into_tensor(gathered, piece)
    return gathered


def calibration_steps(total_batches):
    """Steps every rank must take so that all of them issue the same collectives: ceil(total / world)."""
    w = world_size()
    return (int(total_batches) + w - 1) // w


def shard_loader_kwargs():
    """kwargs for mx.gluon.data.DataLoader: every rank draws the same sampler sequence, keeps batches i % W == r."""
    return {"rank": rank(), "world_size": world_size()}


# ---- naive-EMA calibration ------------------------------------------------------------------------------------------
def _calibrated_blocks(net):
    """Blocks that own a calibration scalar, in net order: convolutions / Dense (`input_max`) and quantised Activations
    (`act_max`) — both kinds of slot live in the net's arena and both must end a step identical on every rank."""
    return [b for b in net.collect_quantized_blocks()
            if getattr(b, "input_max", None) is not None or getattr(b, "act_max", None) is not None]


def _would_exchange(block):
    """Does this block's forward reach its strict-mode exchange (the converters' own early-return conditions)?"""
    args = getattr(block, "quantize_args", None)
    if args is None or not getattr(block, "enable_quantize", True):
        return False
    if hasattr(args, "quantize_act"):
        return bool(args.quantize_act)
    return bool(getattr(args, "quantize_input", False))


class _CalibrationSync(object):
    """What both modes share: the blocks, the (L x max_local_batch) statistic matrix whose row l receives layer l's
    per-sample maxima, and the step's local sample count — taken from the INPUT BATCH by a forward pre-hook on the net
    (not from whatever some block saw last), 0 when this rank ran no forward in the step."""

    def __init__(self, net, blocks, stats, device):
        self.blocks, self.stats, self.device = blocks, stats, device
        self.n_step = 0
        self._handle = net.register_forward_pre_hook(self._note_batch)

    def _note_batch(self, _net, inputs):
        self.n_step = int(inputs[0].shape[0])
        if self.n_step > self.stats.shape[1]:
            raise ValueError("calibration batch of %d samples but attach_calibration_sync was given max_local_batch=%d"
                             % (self.n_step, self.stats.shape[1]))

    def take_step_count(self):
        n, self.n_step = self.n_step, 0
        for b in self.blocks:                      # a block that ran this step saw the same batch as the net
            seen = getattr(b, "_fq_last_n", None)
            if n and seen not in (None, 0, n):
                raise RuntimeError("%s saw %d samples in a step whose input batch has %d: the calibration collective "
                                   "assumes one batch size per step" % (b.name, seen, n))
            b._fq_last_n = 0
        return n

    def detach(self):
        self._handle.detach()


class _LayerCollective(_CalibrationSync):
    """STRICT mode: one all-gather per quantised layer per forward, between its statistic pass and its apply pass."""

    def __init__(self, net, blocks, stats, device):
        super(_LayerCollective, self).__init__(net, blocks, stats, device)
        W = world_size()
        self.pack = torch.zeros(1 + stats.shape[1], dtype=torch.float32, device=device)
        self.gathered = torch.zeros(W * self.pack.numel(), dtype=torch.float32, device=device)
        self.order = []                  # id of every block whose exchange ran in the current forward, in call order
        self.last_order = None
        by_id = {id(b): b for b in blocks}
        self.by_id = by_id
        live = group_is_live()
        for b in blocks:
            b._fq_global_stat = self._hook_for(b) if live else None
            b._fq_keep_rows = False

    def _hook_for(self, block):
        def _hook(per_sample, n_local, out):
            self.order.append(id(block))
            return _global_mean(per_sample, int(n_local), self.pack, self.gathered, out)
        return _hook

    def _note_batch(self, _net, inputs):
        super(_LayerCollective, self)._note_batch(_net, inputs)
        self.order = []

    def __call__(self, net, arena):
        """From `net.update_ema()`.  A rank that ran a forward already holds the global statistic in every slot.  A rank
        WITHOUT a batch in this step issues the same all-gathers now, in the order the forward issues them (the order
        its own last forward recorded; the net's block order before any forward) with an empty record each — so every
        rank takes part in the same 27-53 collectives and ends the step with the same `current_*` values."""
        n = self.take_step_count()
        if n:
            self.last_order = list(self.order)
            return
        if not group_is_live():
            return
        slot = {}
        for j, (blk, _pattr, _cattr, _pub) in enumerate(arena.slots):
            slot[id(blk)] = arena.cur[j:j + 1]
        # before any forward of its own: the blocks whose forward WOULD exchange, by the predicate the converters use
        # (convert_conv2d / convert_dense: enable_quantize and quantize_args.quantize_input; convert_act: enable_quantize and
        # quantize_args.quantize_act) - a disabled block, or an Activation converted with quantize_act=False, owns a slot but
        # issues no all-gather, and replaying one for it would leave the ranks with different collective counts
        order = self.last_order if self.last_order else [id(b) for b in self.blocks if _would_exchange(b)]
        for bid in order:
            _global_mean(None, 0, self.pack, self.gathered, slot[bid])


class _StepCollective(_CalibrationSync):
    """The default mode: ONE all-reduce per calibration step, issued from `net.update_ema()`."""

    def __init__(self, net, blocks, stats, device):
        super(_StepCollective, self).__init__(net, blocks, stats, device)
        self.record = torch.zeros(len(blocks) + 1, dtype=torch.float64, device=device)
        self.means = torch.zeros(len(blocks), dtype=torch.float32, device=device)
        self.slot_index = None
        for b in blocks:
            b._fq_global_stat = None
            b._fq_keep_rows = True           # the converter leaves each layer's per-sample maxima in its row of `stats`

    def _slots(self, arena):
        if self.slot_index is None or self.slot_index.numel() != len(self.blocks):
            where = {id(blk): j for j, (blk, _pattr, _cattr, _pub) in enumerate(arena.slots)}
            self.slot_index = torch.tensor([where[id(b)] for b in self.blocks], dtype=torch.long,
                                           device=self.means.device)
        return self.slot_index

    def __call__(self, net, arena):
        n_local = self.take_step_count()
        ops.stat_rows_sum(self.stats, n_local, out=self.record)
        if group_is_live():      # (also with one rank: the collective is still issued)
            all_reduce(self.record)
        ops.mean_from_sums(self.record, out=self.means)
        arena.cur.index_copy_(0, self._slots(arena), self.means)


def _global_mean(per_sample, n_local, pack, gathered, out):
    """All-gather one layer's per-sample maxima (+ the local count) and form the GLOBAL batch mean in global sample
    order (rank-major) with the same ordered fp64 mean as the single-device kernel.  One collective, no host sync:
    the (possibly ragged) per-rank counts travel inside the records and are consumed on the device."""
    pack.zero_()
    pack[0] = float(n_local)
    if n_local:
        pack[1:1 + n_local].copy_(per_sample[:n_local])
    all_gather_into(gathered, pack)
    ops.batch_mean_gathered(gathered.view(world_size(), -1), out=out)
    return out


def attach_calibration_sync(net, max_local_batch, strict=False):
    """Make every rank end each calibration step with the statistics of the GLOBAL batch (module docstring: default =
    one all-reduce per step; strict = per-layer all-gather, bit-identical to one device that sees the global batch)."""
    detach_calibration_sync(net)
    blocks = _calibrated_blocks(net)
    first = blocks[0]
    device = (first.input_max if getattr(first, "input_max", None) is not None else first.act_max).data()._t.device
    stats = torch.zeros(len(blocks), int(max_local_batch), dtype=torch.float32, device=device)
    for i, b in enumerate(blocks):
        b._fq_stat_ws = stats[i]
        b._fq_last_n = 0
    net._fq_stat_matrix = stats
    net._fq_calibration_sync = (_LayerCollective if strict else _StepCollective)(net, blocks, stats, device)
    return net


def empty_calibration_step(net, momentum=0.9):
    """A rank whose shard has no batch in this step still takes part in the step's collective(s) — an empty record per
    collective, in both modes — and applies the same EMA update as the others."""
    sync = getattr(net, "_fq_calibration_sync", None)
    if sync is not None:
        sync.n_step = 0
    net.update_ema(momentum)


def detach_calibration_sync(net):
    sync = getattr(net, "_fq_calibration_sync", None)
    if sync is not None:
        sync.detach()
    for b in net.collect_quantized_blocks():
        b._fq_global_stat = None
        b._fq_keep_rows = False
        b._fq_stat_ws = None
    net._fq_calibration_sync = None
    return net


# ---- KL calibration ----------------------------------------------------------------------------------------------------
def kl_sync(stage, tensor):
    """`sync` hook for quantize.distribution_calibrate.collect_feature_maps."""
    if not group_is_live():                      # (a one-rank group still issues them: the transport is what is exercised)
        return
    if stage == "range":
        broadcast(tensor, src=0)
    elif stage == "hist":
        all_reduce(tensor)
    else:
        raise ValueError(stage)


# ---- evaluation ------------------------------------------------------------------------------------------------------------
def allreduce_eval_counters(counters):
    """[test_num_correct, total, correct_counter[classes], label_counter[classes]] summed over ranks."""
    if group_is_live():
        all_reduce(counters)
    return counters
