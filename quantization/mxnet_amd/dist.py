"""One process per GPU: batch sharding and the (tiny) collectives of calibration and evaluation.

The reference is single-process, single-device (no kvstore / NCCL / multi-ctx split anywhere), so
everything here is additive and defined by ONE requirement: N ranks that each see 1/N of a batch must end with the
numbers one device would have produced on the whole batch.  The path shards over independent images; exchanges are:

  evaluation        one all-reduce(sum) of the accuracy counters at the end (simulate_quantization.py:123-147)
  online eval       none: each rank's local batch is "the batch" (== the reference run with --batch-size=local)
  naive-EMA calib   per quantised layer ONE all-gather of its (1 + n_local) per-sample maxima, between the layer's
                    statistic pass and its apply pass -> every rank applies the GLOBAL batch mean (global sample order,
                    same ordered fp64 mean as the single-device kernel) and later runs the identical EMA update:
                    replicas stay bit-identical to each other and to one device that saw the global batch
  KL calib          all-reduce(max) of the first-batch ranges `fm_max[L]`, and ONE all-reduce(sum) of the exact int64
                    histograms [L x 2048] (434 KB for ResNet-50) at the end (distribution_calibrate.py:97-104)

Transport: `torch.distributed` — backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.  All messages
are latency-bound (<= 434 KB), so they go on the compute stream with RCCL's defaults; no bucketing is needed.
"""
import os

import torch
import torch.distributed as dist

from . import ops

__all__ = ["init", "is_distributed", "rank", "world_size", "shard_loader_kwargs", "attach_calibration_sync", "detach_calibration_sync",
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


def attach_calibration_sync(net, max_local_batch):
    """Make calibration under batch sharding reproduce ONE device that sees the global batch, bit for bit.

    During naive calibration the reference quantises each layer's input ONLINE with the statistic of the current batch
    (simulate_quantization.py:322) and that statistic is what `update_ema` consumes, so a layer's output depends on its
    batch-mates: the global statistic is needed BEFORE the layer's apply pass, not just before the EMA.  Each quantised
    block therefore gets a hook (`_fq_global_stat`) that the converter calls between its statistic pass and its apply
    pass: one all-gather of (1 + n_local) floats per layer — latency-bound, ~27-53 small collectives per forward.
    (Online EVALUATION does not install this: there each rank's local batch is "the batch", which equals the reference
    run with --batch-size=local.)"""
    blocks = [b for b in net.collect_quantized_blocks() if getattr(b, "input_max", None) is not None]
    device = blocks[0].input_max.data()._t.device
    W = world_size()
    stats = torch.zeros(len(blocks), int(max_local_batch), dtype=torch.float32, device=device)
    pack = torch.zeros(1 + int(max_local_batch), dtype=torch.float32, device=device)
    gathered = torch.zeros(W * pack.numel(), dtype=torch.float32, device=device)

    def _hook(per_sample, n_local, out):
        return _global_mean(per_sample, int(n_local), pack, gathered, out)

    for i, b in enumerate(blocks):
        b._fq_stat_ws = stats[i]
        b._fq_last_n = 0
        b._fq_global_stat = _hook if W > 1 else None
    net._fq_stat_matrix = stats
    net._fq_calibration_sync = None          # current_input_max already is the global statistic when update_ema runs
    return net


def detach_calibration_sync(net):
    for b in net.collect_quantized_blocks():
        b._fq_global_stat = None
    return net


# ---- KL calibration ----------------------------------------------------------------------------------------------------
def kl_sync(stage, tensor):
    """`sync` hook for quantize.distribution_calibrate.collect_feature_maps."""
    if not is_distributed():
        return
    if stage == "max":
        dist.all_reduce(tensor, op=dist.ReduceOp.MAX)
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
