"""The N > 1 path on CPU: two gloo ranks (torch.distributed.run, 127.0.0.1) against one process that sees the whole
batch.  Claim under test (dist.py): sharding the batch changes NOTHING — EMA thresholds, KL histograms / ranges and
eval counters are bit-identical to the single-device run on the global batch, and identical across ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle.patch import oracle_ops


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_process(local_bs, world, ragged):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker as W
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps
    with oracle_ops():
        net = W.make_net()
        net.quantize_input(enable=True, online=True)
        blocks = net.collect_quantized_blocks()
        ema = []
        for step, glob in enumerate(W.batches(4, local_bs * world)):
            if ragged and step == 3:
                glob = glob[:local_bs + 1]
            net(mx.nd.array(glob))
            net.update_ema()
            ema.append([b.input_max.data().asscalar() for b in blocks])
        net2 = W.make_net()
        net2.disable_quantize()
        b2 = net2.collect_quantized_blocks()
        # single device: the first batch alone fixes the range; two ranks fix it with max over THEIR first batches
        # (batches 0 and 1), so the single-device equivalent is a first "batch" made of both.
        all_b = W.batches(4, local_bs, seed=21)
        first = np.concatenate(all_b[:world])
        loader = [(mx.nd.array(first), None)] + [(mx.nd.array(b), None) for b in all_b[world:]]
        hists, maxes = collect_feature_maps(net2, 64, loader, mx.cpu())
    return (np.asarray(ema, np.float32), np.stack([hists[b] for b in b2]),
            np.asarray([maxes[b] for b in b2], np.float32))


@pytest.mark.parametrize("ragged", [0, 1])
def test_two_ranks_equal_one_device_on_the_global_batch(tmp_path, ragged):
    local_bs, world = 3, 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(local_bs), str(ragged)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-3000:]
    r = [np.load(os.path.join(tmp_path, "rank%d.npz" % i)) for i in range(world)]
    for k in ("ema", "hist", "fm_max", "counters"):
        np.testing.assert_array_equal(r[0][k], r[1][k], "ranks disagree on " + k)
    ema, hist, fm_max = _single_process(local_bs, world, ragged)
    np.testing.assert_array_equal(r[0]["ema"], ema, "EMA thresholds differ from the single-device global-batch run")
    np.testing.assert_array_equal(r[0]["fm_max"], fm_max)
    np.testing.assert_array_equal(r[0]["hist"], hist)
    np.testing.assert_array_equal(r[0]["counters"], [3.0, 20.0])


def test_loader_shards_batches_round_robin():
    from quantization.mxnet_amd.mx.gluon.data import DataLoader, ArrayDataset
    ds = ArrayDataset(np.arange(23, dtype=np.float32), np.arange(23, dtype=np.int64))
    seen = []
    for r in range(3):
        dl = DataLoader(ds, batch_size=4, last_batch="keep", rank=r, world_size=3)
        got = [x.asnumpy() for x, _ in dl]
        assert len(got) == len(dl)
        seen.append(got)
    flat = sorted(float(v) for g in seen for b in g for v in b)
    assert flat == list(map(float, range(23)))
    assert [len(g) for g in seen] == [2, 2, 2]
    assert seen[2][-1].tolist() == [20.0, 21.0, 22.0]          # ragged tail kept (last_batch='keep')
