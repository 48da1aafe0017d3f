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
"""
import os

import torch
import torch.distributed as dist

from . import ops

__all__ = ["init", "is_distributed", "rank", "world_size", "shard_loader_kwargs", "attach_calibration_sync", "detach_calibration_sync", "empty_calibration_step", "calibration_steps",
           "kl_sync", "allreduce_eval_counters", "shutdown"]


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def init(backend=None):
    """Join the job described by torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world).  A no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, int(os.environ.get("LOCAL_RANK", "0")), 1
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return dist.get_rank(), local, dist.get_world_size()


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def calibration_steps(total_batches):
    """Steps every rank must take so that all of them issue the same collectives: ceil(total / world)."""
    w = world_size()
    return (int(total_batches) + w - 1) // w


def shard_loader_kwargs():
    """kwargs for mx.gluon.data.DataLoader: every rank draws the same sampler sequence, keeps batches i % W == r."""
    return {"rank": rank(), "world_size": world_size()}


# ---- naive-EMA calibration ------------------------------------------------------------------------------------------
def _global_mean(per_sample, n_local, pack, gathered, out):
    """All-gather one layer's per-sample maxima (+ the local count) and form the GLOBAL batch mean in global sample
    order (rank-major) with the same ordered fp64 mean as the single-device kernel.  One collective, no host sync:
    the (possibly ragged) per-rank counts travel inside the records and are consumed on the device."""
    pack.zero_()
    pack[0] = float(n_local)
    pack[1:1 + n_local].copy_(per_sample[:n_local])
    dist.all_gather_into_tensor(gathered, pack)
    ops.batch_mean_gathered(gathered.view(world_size(), -1), out=out)
    return out


def _strict_attach(net, blocks, stats, device, max_local_batch):
    """Per-layer exchange between the statistic pass and the apply pass (see the module docstring)."""
    W = world_size()
    pack = torch.zeros(1 + int(max_local_batch), dtype=torch.float32, device=device)
    gathered = torch.zeros(W * pack.numel(), dtype=torch.float32, device=device)

    def _hook(per_sample, n_local, out):
        return _global_mean(per_sample, int(n_local), pack, gathered, out)

    for b in blocks:
        b._fq_global_stat = _hook if (dist.is_available() and dist.is_initialized()) else None
        b._fq_keep_rows = False
    net._fq_calibration_sync = None          # current_input_max already is the global statistic when update_ema runs


class _StepCollective(object):
    """The default mode: ONE all-reduce per calibration step, issued from `net.update_ema()`."""

    def __init__(self, net, blocks, stats, device):
        self.blocks, self.stats = blocks, stats
        self.record = torch.zeros(len(blocks) + 1, dtype=torch.float64, device=device)
        self.means = torch.zeros(len(blocks), dtype=torch.float32, device=device)
        self.slot_index = None
        self.skip = False                    # set by empty_calibration_step: this rank saw no batch in this step

    def _slots(self, arena):
        if self.slot_index is None or self.slot_index.numel() != len(self.blocks):
            where = {}
            for j, (blk, pattr, _, _) in enumerate(arena.slots):
                if pattr == "input_max":
                    where[id(blk)] = j
            self.slot_index = torch.tensor([where[id(b)] for b in self.blocks], dtype=torch.long,
                                           device=self.means.device)
        return self.slot_index

    def __call__(self, net, arena):
        n_local = 0 if self.skip else int(self.blocks[0]._fq_last_n)
        self.skip = False
        ops.stat_rows_sum(self.stats, n_local, out=self.record)
        if dist.is_available() and dist.is_initialized():      # (also with one rank: the collective is still issued)
            dist.all_reduce(self.record, op=dist.ReduceOp.SUM)
        ops.mean_from_sums(self.record, out=self.means)
        arena.cur.index_copy_(0, self._slots(arena), self.means)


def attach_calibration_sync(net, max_local_batch, strict=False):
    """Make every rank end each calibration step with the statistics of the GLOBAL batch (module docstring: default =
    one all-reduce per step; strict = per-layer all-gather, bit-identical to one device that sees the global batch)."""
    blocks = [b for b in net.collect_quantized_blocks() if getattr(b, "input_max", None) is not None]
    device = blocks[0].input_max.data()._t.device
    stats = torch.zeros(len(blocks), int(max_local_batch), dtype=torch.float32, device=device)
    for i, b in enumerate(blocks):
        b._fq_stat_ws = stats[i]
        b._fq_last_n = 0
    net._fq_stat_matrix = stats
    if strict:
        _strict_attach(net, blocks, stats, device, max_local_batch)
    else:
        for b in blocks:
            b._fq_global_stat = None
            b._fq_keep_rows = True           # the converter leaves each layer's per-sample maxima in its row of `stats`
        net._fq_calibration_sync = _StepCollective(net, blocks, stats, device)
    return net


def empty_calibration_step(net, momentum=0.9):
    """A rank whose shard has no batch in this step still takes part in the step's collective (an empty record) and
    applies the same EMA update as the others."""
    sync = getattr(net, "_fq_calibration_sync", None)
    if isinstance(sync, _StepCollective):
        sync.skip = True
    net.update_ema(momentum)


def detach_calibration_sync(net):
    for b in net.collect_quantized_blocks():
        b._fq_global_stat = None
        b._fq_keep_rows = False
        b._fq_stat_ws = None
    net._fq_calibration_sync = None
    return net


# ---- KL calibration ----------------------------------------------------------------------------------------------------
def kl_sync(stage, tensor):
    """`sync` hook for quantize.distribution_calibrate.collect_feature_maps."""
    if not is_distributed():
        return
    if stage == "range":
        dist.broadcast(tensor, src=0)
    elif stage == "hist":
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    else:
        raise ValueError(stage)


# ---- evaluation ------------------------------------------------------------------------------------------------------------
def allreduce_eval_counters(counters):
    """[test_num_correct, total, correct_counter[classes], label_counter[classes]] summed over ranks."""
    if is_distributed():
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    return counters
