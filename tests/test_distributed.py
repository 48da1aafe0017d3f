"""The N > 1 path on CPU: two gloo ranks (torch.distributed.run, 127.0.0.1) against one process.

Claims under test (dist.py):
  * strict mode: sharding the batch changes NOTHING — EMA thresholds are bit-identical to the single-device run on the
    global batch (ragged tail included);
  * default mode (ONE all-reduce per calibration step): every rank ends each step with the batch mean of the GLOBAL batch
    over the per-sample maxima the ranks computed — checked against a recomputation from the ranks' own rows — and the
    first quantised layer (whose input does not depend on batch-mates) equals the single-device run bit for bit; ranks
    that have no batch in a step (odd batch counts) still take part and do not hang;
  * KL: ranges = the first global batch's (broadcast from rank 0), histograms summed exactly: bit-identical to one device
    walking the same batches in the same order, also when a rank's shard is empty;
  * evaluation counters: one all-reduce."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import fq_oracle as O
from oracle.patch import oracle_ops

sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_process(local_bs, world, case):
    import dist_worker as W
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps
    with oracle_ops():
        net = W.make_net(with_act="_act" in case, disable_one="_off" in case)
        net.quantize_input(enable=True, online=True)
        ema = []
        for shards in W.calib_steps(case, local_bs, world):
            glob = np.concatenate([s for s in shards if s is not None and len(s)])
            net(mx.nd.array(glob))
            net.update_ema()
            ema.append(W.calibration_scalars(net))
        net2 = W.make_net()
        net2.disable_quantize()
        b2 = net2.collect_quantized_blocks()
        loader = [(mx.nd.array(b), None) for b in W.kl_batches(case, local_bs)]      # the same batches, the same order
        hists, maxes = collect_feature_maps(net2, 64, loader, mx.cpu())
    return (np.asarray(ema, np.float32), np.stack([hists[b] for b in b2]),
            np.asarray([maxes[b] for b in b2], np.float32))


def _run_two_ranks(tmp_path, local_bs, case):
    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(local_bs), case]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-3000:]
    r = [np.load(os.path.join(tmp_path, "rank%d.npz" % i)) for i in range(world)]
    for k in ("ema", "hist", "fm_max", "counters"):
        np.testing.assert_array_equal(r[0][k], r[1][k], "ranks disagree on " + k)
    np.testing.assert_array_equal(r[0]["counters"], [3.0, 20.0])
    return r


@pytest.mark.parametrize("case", ["strict", "strict_ragged", "strict_short", "strict_act", "strict_off_firstempty"])
def test_strict_mode_equals_one_device_on_the_global_batch(tmp_path, case):
    local_bs, world = 3, 2
    r = _run_two_ranks(tmp_path, local_bs, case)
    ema, hist, fm_max = _single_process(local_bs, world, case)
    np.testing.assert_array_equal(r[0]["ema"], ema, "EMA thresholds differ from the single-device global-batch run")
    np.testing.assert_array_equal(r[0]["fm_max"], fm_max)
    np.testing.assert_array_equal(r[0]["hist"], hist)


@pytest.mark.parametrize("case", ["step", "step_ragged", "step_short", "step_act"])
def test_one_collective_per_step_gives_the_ema_of_the_global_batch_mean(tmp_path, case):
    local_bs, world = 3, 2
    r = _run_two_ranks(tmp_path, local_bs, case)
    steps, layers = r[0]["rows"].shape[0], r[0]["rows"].shape[1]
    # expected: per step and layer the ordered batch mean over rank 0's samples then rank 1's, then the reference's EMA
    state = np.zeros(layers, np.float32)
    for s in range(steps):
        cur = np.zeros(layers, np.float32)
        for l in range(layers):
            vals = np.concatenate([r[k]["rows"][s, l, :int(r[k]["counts"][s])] for k in range(world)])
            cur[l] = O.batch_mean(vals)
        state = O.ema_update(state, cur, 0.9)
        np.testing.assert_array_equal(r[0]["ema"][s], state, "step %d" % s)
    ema, hist, fm_max = _single_process(local_bs, world, case)
    np.testing.assert_array_equal(r[0]["ema"][:, 0], ema[:, 0], "first quantised layer differs from one device")
    assert np.allclose(r[0]["ema"], ema, rtol=0.2), "local-batch forward drifted far from the global-batch run"
    np.testing.assert_array_equal(r[0]["fm_max"], fm_max)
    np.testing.assert_array_equal(r[0]["hist"], hist)


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
