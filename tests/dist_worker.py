"""Worker for tests/test_distributed.py: launched by torch.distributed.run with world_size 2 on CPU (gloo).
Runs the product's calibration host logic (EMA sync, KL sync, eval counters) with the oracle standing in for the HIP
entry points, and writes what each rank ended with to <out>/rank<r>.npz."""
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
from quantization.mxnet_amd.mx.gluon import nn  # noqa: E402
from quantization.mxnet_amd.quantize import convert  # noqa: E402
from quantization.mxnet_amd.quantize.initialize import qparams_init  # noqa: E402


def make_net(seed=3):
    rng = np.random.default_rng(seed)
    shapes = {"tiny_conv0_weight": (8, 3, 3, 3), "tiny_conv1_weight": (8, 1, 3, 3), "tiny_conv1_bias": (8,),
              "tiny_conv2_weight": (12, 8, 1, 1), "tiny_dense0_weight": (5, 12), "tiny_dense0_bias": (5,)}
    params = {k: (rng.standard_normal(s) * 0.4).astype(np.float32) for k, s in shapes.items()}
    net = tiny_net(params)
    convert.convert_model(net, exclude=[net[0]])
    qparams_init(net)
    return net


def batches(n_batches, bs, seed=9):
    rng = np.random.default_rng(seed)
    return [(rng.standard_normal((bs, 3, 8, 8)) * (1 + 0.25 * i)).astype(np.float32) for i in range(n_batches)]


def main():
    out_dir, local_bs, ragged = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, _, world = fqdist.init("gloo")
    with oracle_ops():
        # ---- naive-EMA calibration on the rank's shard of each global batch -------------------------------------
        net = make_net()
        fqdist.attach_calibration_sync(net, local_bs)
        net.quantize_input(enable=True, online=True)
        blocks = net.collect_quantized_blocks()
        ema = []
        for step, glob in enumerate(batches(4, local_bs * world)):
            if ragged and step == 3:
                glob = glob[:local_bs + 1]                 # last global batch: rank 0 full, rank 1 one sample
            mine = glob[rank * local_bs:(rank + 1) * local_bs]
            net(mx.nd.array(mine))
            net.update_ema()
            ema.append([b.input_max.data().asscalar() for b in blocks])
        # ---- KL collection ------------------------------------------------------------------------------------------
        net2 = make_net()
        net2.disable_quantize()
        all_b = batches(4, local_bs, seed=21)
        loader = [(mx.nd.array(b), None) for i, b in enumerate(all_b) if i % world == rank]
        hists, maxes = collect_feature_maps(net2, 64, loader, mx.cpu(), sync=fqdist.kl_sync)
        b2 = net2.collect_quantized_blocks()
        counters = torch.tensor([float(rank + 1), 10.0])
        fqdist.allreduce_eval_counters(counters)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ema=np.asarray(ema, np.float32),
             hist=np.stack([hists[b] for b in b2]), fm_max=np.asarray([maxes[b] for b in b2], np.float32),
             counters=counters.numpy())
    fqdist.shutdown()


if __name__ == "__main__":
    main()
