"""Worker for tests/test_distributed.py: launched by torch.distributed.run with world_size 2 on CPU (gloo).
Runs the product's calibration host logic (both EMA collective modes, KL sync, eval counters) with the oracle standing in
for the HIP entry points, and writes what each rank ended with to <out>/rank<r>.npz.

    dist_worker.py <out_dir> <local_bs> <case>      case: strict | strict_ragged | strict_short | strict_act | strict_off_firstempty | step | step_ragged | step_short | step_act
(`*_act`: the net's Activations are converted too — their `act_max` slots take part in the same collectives)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.patch import oracle_ops  # noqa: E402
from quantization.mxnet_amd import mx, dist as fqdist  # noqa: E402
from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps  # noqa: E402
from test_host_logic import tiny_net  # noqa: E402
from quantization.mxnet_amd.quantize import convert  # noqa: E402
from quantization.mxnet_amd.quantize.initialize import qparams_init  # noqa: E402


def make_net(seed=3, with_act=False, disable_one=False):
    rng = np.random.default_rng(seed)
    shapes = {"tiny_conv0_weight": (8, 3, 3, 3), "tiny_conv1_weight": (8, 1, 3, 3), "tiny_conv1_bias": (8,),
              "tiny_conv2_weight": (12, 8, 1, 1), "tiny_dense0_weight": (5, 12), "tiny_dense0_bias": (5,)}
    params = {k: (rng.standard_normal(s) * 0.4).astype(np.float32) for k, s in shapes.items()}
    net = tiny_net(params)
    if with_act:
        from quantization.mxnet_amd.mx.gluon import nn
        fns = dict(convert.default_convert_fn)
        fns[nn.Activation] = convert.gen_act_converter()
        convert.convert_model(net, exclude=[net[0], net[1]], convert_fn=fns)
    else:
        convert.convert_model(net, exclude=[net[0]])
    qparams_init(net)
    if disable_one:          # a block switched off owns a calibration slot but its forward never reaches the exchange
        net.collect_quantized_blocks()[1].enable_quantize = False
    return net


def calibration_scalars(net):
    """Every calibration scalar of the net in block order: input_max of conv / Dense blocks, act_max of Activations."""
    out = []
    for b in net.collect_quantized_blocks():
        p = getattr(b, "input_max", None)
        if p is None:
            p = getattr(b, "act_max", None)
        if p is not None:
            out.append(p.data().asscalar())
    return out


def batches(n_batches, bs, seed=9):
    rng = np.random.default_rng(seed)
    return [(rng.standard_normal((bs, 3, 8, 8)) * (1 + 0.25 * i)).astype(np.float32) for i in range(n_batches)]


def calib_steps(case, local_bs, world):
    """The global batches of the calibration, as the list of per-rank shards of every step (None = no batch)."""
    steps = []
    glob = batches(4, local_bs * world)
    for step, g in enumerate(glob):
        if case.endswith("_ragged") and step == 3:
            g = g[:local_bs + 1]                           # last global batch: rank 0 full, rank 1 one sample
        shards = [g[r * local_bs:(r + 1) * local_bs] for r in range(world)]
        if case.endswith("_short") and step == 3:
            shards = [g[:local_bs]] + [None] * (world - 1)  # odd batch count: only rank 0 has a batch in the last step
        if case.endswith("_firstempty") and step == 0:
            shards = [g[:local_bs]] + [None] * (world - 1)  # the other ranks' shards are empty FROM STEP 0: no forward yet
        steps.append(shards)
    return steps


def kl_batches(case, local_bs):
    return batches(1 if case.endswith("_short") else (3 if case.endswith("_ragged") else 4), local_bs, seed=21)


def main():
    out_dir, local_bs, case = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, _, world = fqdist.init("gloo")
    with oracle_ops():
        # ---- naive-EMA calibration on the rank's shard of each global batch -------------------------------------
        net = make_net(with_act="_act" in case, disable_one="_off" in case)
        fqdist.attach_calibration_sync(net, local_bs, strict=case.startswith("strict"))
        net.quantize_input(enable=True, online=True)
        blocks = [b for b in net.collect_quantized_blocks() if hasattr(b, "_fq_stat_ws")]
        ema, rows = [], []
        for shards in calib_steps(case, local_bs, world):
            mine = shards[rank]
            if mine is None or len(mine) == 0:
                fqdist.empty_calibration_step(net)
                rows.append(np.zeros((len(blocks), 0), np.float32))
            else:
                net(mx.nd.array(mine))
                rows.append(net._fq_stat_matrix[:, :len(mine)].numpy().copy())
                net.update_ema()
            ema.append(calibration_scalars(net))
        # ---- KL collection: batches strided over the ranks like the loader does -----------------------------------
        net2 = make_net()
        net2.disable_quantize()
        loader = [(mx.nd.array(b), None) for i, b in enumerate(kl_batches(case, local_bs)) if i % world == rank]
        hists, maxes = collect_feature_maps(net2, 64, loader, mx.cpu(), sync=fqdist.kl_sync)
        b2 = net2.collect_quantized_blocks()
        counters = torch.tensor([float(rank + 1), 10.0])
        fqdist.allreduce_eval_counters(counters)
    rows_padded = np.zeros((len(rows), len(blocks), local_bs), np.float32)
    counts = np.zeros(len(rows), np.int64)
    for i, r in enumerate(rows):
        rows_padded[i, :, :r.shape[1]] = r
        counts[i] = r.shape[1]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ema=np.asarray(ema, np.float32), rows=rows_padded,
             counts=counts, hist=np.stack([hists[b] for b in b2]),
             fm_max=np.asarray([maxes[b] for b in b2], np.float32), counters=counters.numpy())
    fqdist.shutdown()


if __name__ == "__main__":
    main()
