"""The N > 1 paths through the REAL kernels on a one-GPU box: two ranks that share GPU 0 and talk over gloo (RCCL refuses
two ranks on one device - profiles/r2_rccl_smoke.txt; FQ_*_BACKEND=gloo stages device tensors through the host around each
collective, everything else is the production code).

  * `bench.py --gpus 2` typed WITHOUT a launcher: the parent starts its own ranks before touching a GPU, passes rank 0's
    JSON line through and exits with the children's status (what the driver's multi-GPU bench would type);
  * the CLI's calibration flows (examples/simulate_quantization.py) - naive EMA in both collective modes and KL - on two
    ranks: thresholds and accuracies equal ONE process bit for bit (strict mode: one device on the global batch; KL: one
    device walking the same batches; default mode: the EMA of the global batch means recomputed from one process'
    per-sample statistics), including a rank WITHOUT a batch in the last step."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import fq_oracle as O

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("streams", [2, 1])
def test_bench_launches_its_own_ranks(gpu, streams):
    env = dict(os.environ, FQ_BENCH_SHARE_GPU="1", FQ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--max-repeats", "4", "--streams", str(streams)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 2
    assert rec["config"]["global_batch"] == 256
    assert rec["scaling"] == "weak" and rec["unit"] == "images/sec" and rec["value"] > 0
    assert rec["repeats"] == 4 and rec["consistency"]["blocks"] == 4
    # counters summed over BOTH ranks in one all-reduce: every step either rank ran (warm-up + 4 blocks of 3) x 128 images;
    # with two steps in flight also the set-up forward of each stream and the one-stream block of 3 steps; with hipGraph replay
    # the first replay of every captured graph (streams x 4 resident batches)
    extra = (2 + 3) if streams == 2 else 0
    assert rec["config"]["hipgraph"] is True and rec["config"]["hipgraph_error"] is None
    extra += streams * 4
    assert rec["eval_counters"]["images"] == 2 * 128 * (2 + 4 * 3 + extra)
    assert rec["config"]["streams"] == streams and ("single_stream" in rec) == (streams == 2)
    assert abs(rec["value"] - 256 * 3 / (rec["ms_per_step"] * 3e-3)) / rec["value"] < 1e-3
    assert rec["roofline"]["frac"] > 0 and "cpu_baseline" not in rec and "headline_tensor" not in rec
    # what a first multi-GPU line is checked with in one pass: the transport, each rank's own clock, the data-path collectives
    # (an evaluation step has none), and no HBM-traffic figure that this run did not measure
    rk = rec["ranks"]
    assert rk["world"] == 2 and rk["backend"] == "gloo" and rk["rccl_world"] is None
    assert 0 < rk["ms_per_step_min_over_ranks"] <= rk["ms_per_step_max_over_ranks"] <= rec["consistency"]["ms_per_step_max"] * 1.001
    assert rk["collectives_in_timed_region"] == {} and rk["collectives_per_step"] == {}
    assert rec["roofline"]["traffic"] is None and rec["roofline"]["frac_actual"] == rec["roofline"]["frac"]


@pytest.mark.parametrize("phase", ["calib-naive", "calib-kl"])
def test_bench_calibration_phases_report_their_collectives(gpu, phase):
    """`bench.py --gpus 2 --phase calib-naive|calib-kl`: the line says how many data-path collectives a step issued and how
    large they were - ONE all-reduce of L + 1 doubles per naive-EMA step (north_star's collective); one range broadcast and one
    exact-histogram all-reduce per KL collection."""
    env = dict(os.environ, FQ_BENCH_SHARE_GPU="1", FQ_BENCH_BACKEND="gloo", FQ_DIST_BACKEND="gloo", FQ_DIST_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--phase", phase,
           "--model", "cifar_resnet20_v1", "--batch-size", "16", "--no-cpu-baseline", "--max-repeats", "2"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    rk = rec["ranks"]
    assert rk["world"] == 2 and rec["repeats"] == 2
    if phase == "calib-naive":
        assert set(rk["collectives_per_step"]) == {"all_reduce"} and rk["collectives_per_step"]["all_reduce"] == 1.0
        # L + 1 doubles: the per-layer sums and the sample count of the global batch
        assert rk["bytes_per_collective"]["all_reduce"] % 8 == 0 and 8 * 10 < rk["bytes_per_collective"]["all_reduce"] < 8 * 64
    else:
        blocks = rec["repeats"]
        assert rk["collectives_in_timed_region"]["broadcast"]["calls"] == blocks          # the first batch's ranges
        assert rk["collectives_in_timed_region"]["all_reduce"]["calls"] == blocks         # the histograms, once per collection
        assert rk["bytes_per_collective"]["all_reduce"] % (2048 * 8) == 0                 # [L x 2048] int64


@pytest.mark.parametrize("phase", ["eval", "calib-naive", "calib-kl"])
def test_eight_ranks_sharing_gpu0_through_every_phase(gpu, phase):
    """The world size the driver will use, on the one GPU this box has: eight processes on device 0 over
    gloo through the production step of each phase.  Hang-freedom (the timeout), rank 0's line, the collectives per step and -
    for the calibration phases - the SAME thresholds on all eight ranks, bit for bit (`thresholds_equal_on_all_ranks`)."""
    env = dict(os.environ, FQ_BENCH_SHARE_GPU="1", FQ_BENCH_BACKEND="gloo", FQ_DIST_BACKEND="gloo", FQ_DIST_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--phase", phase,
           "--model", "cifar_resnet20_v1", "--batch-size", "8", "--no-cpu-baseline", "--no-headline", "--max-repeats", "2",
           "--min-region-s", "0"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    rk = rec["ranks"]
    assert rec["n_gpus"] == 8 and rk["world"] == 8 and rec["config"]["global_batch"] == 64
    assert 0 < rk["ms_per_step_min_over_ranks"] <= rk["ms_per_step_max_over_ranks"]
    if phase == "eval":
        assert rk["collectives_per_step"] == {} and rk["thresholds_equal_on_all_ranks"] is None
        assert rec["eval_counters"]["images"] > 0 and rec["eval_counters"]["images"] % (8 * 8) == 0
    elif phase == "calib-naive":
        assert rk["collectives_per_step"] == {"all_reduce": 1.0}             # north_star's ONE collective per calibration step
        assert rk["thresholds_equal_on_all_ranks"] is True
    else:
        blocks = rec["repeats"]
        assert rk["collectives_in_timed_region"]["broadcast"]["calls"] == blocks
        assert rk["collectives_in_timed_region"]["all_reduce"]["calls"] == blocks
        assert rk["thresholds_equal_on_all_ranks"] is True


def test_bench_refuses_more_ranks_than_devices(gpu):
    import torch
    want = torch.cuda.device_count() + 1
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("FQ_BENCH_SHARE_GPU", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "GPU(s)" in res.stderr


def _two_ranks(tmp_path, flow):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FQ_DIST_BACKEND="gloo", FQ_DIST_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "gpu_cli_worker.py"), str(tmp_path), flow]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-4000:]
    r = [np.load(os.path.join(tmp_path, "rank%d.npz" % i)) for i in range(2)]
    assert int(r[0]["world"]) == 2 and int(r[0]["device"]) == int(r[1]["device"]) == 0
    np.testing.assert_array_equal(r[0]["thr"], r[1]["thr"], "the two ranks ended calibration with different thresholds")
    assert float(r[0]["acc"]) == float(r[1]["acc"]) and float(r[0]["avg"]) == float(r[1]["avg"])
    return r[0]


def test_cli_naive_calibration_strict_two_ranks_equals_one_device(gpu, tmp_path):
    """--strict-global-batch: thresholds AND the offline evaluation equal one device that sees the global batches."""
    import gpu_cli_worker as W
    r = _two_ranks(tmp_path, "naive_strict")
    cli = W.cli_module()
    thr, acc, avg, _ = W.run_flow(cli, "naive_strict", gpu, 0, 1, W.global_batches("calib", 2), W.global_batches("eval", 2),
                                  2 * W.LOCAL_BS)
    assert np.all(thr > 0)
    np.testing.assert_array_equal(r["thr"], thr)
    assert float(r["acc"]) == acc and float(r["avg"]) == avg


def test_cli_evaluation_three_lanes_ragged_last_batch_two_ranks_equals_one_device(gpu, tmp_path):
    """two ranks, each with THREE evaluation batches in flight replayed from hipGraphs, an odd number of
    batches (rank 1 gets one less) and a ragged last batch on rank 0 (launched eagerly): the counters, summed over the ranks
    in one all-reduce, give the accuracy of one device that evaluates the same images one global batch at a time."""
    import gpu_cli_worker as W
    r = _two_ranks(tmp_path, "strict_lanes")
    assert int(r["replayed"]) >= 2                                     # (rank 0: 6 batches, three of them after a lane's first)
    cli = W.cli_module()
    thr, acc, avg, _ = W.run_flow(cli, "naive_strict", gpu, 0, 1, W.global_batches("calib", 2),
                                  W.global_batches("eval_lanes", 2), 2 * W.LOCAL_BS,
                                  extra=("--eval-streams", "1", "--eval-graph", "0"))
    np.testing.assert_array_equal(r["thr"], thr)
    assert float(r["acc"]) == acc and float(r["avg"]) == avg


def test_cli_naive_calibration_one_collective_per_step_two_ranks(gpu, tmp_path):
    """Default mode: every rank's forward runs on its local batch; the step's ONE all-reduce gives the batch mean of the
    global batch.  Expected values: one process runs the same local batches, the per-sample statistics of every layer are
    read back, and the ordered mean + the reference's EMA are formed on the host by the oracle."""
    import torch
    import gpu_cli_worker as W
    from quantization.mxnet_amd import mx, dist as fqdist
    r = _two_ranks(tmp_path, "naive_step")
    cli = W.cli_module()
    opt = W.options(cli, "naive_step")
    sim = cli.Simulation(opt, gpu, 0, 1)
    np.random.seed(opt.fixed_random_seed)
    sim.build_net()
    sim.quantise_net()
    net = sim.net
    fqdist.attach_calibration_sync(net, W.LOCAL_BS)                 # one process: rows are kept, no collective
    net.quantize_input(enable=True, online=True)
    loc = W.local_batches("calib")
    layers = net._fq_stat_matrix.shape[0]
    state = np.zeros(layers, np.float32)
    for s in range(0, len(loc), 2):
        rows = []
        for x, _ in loc[s:s + 2]:                                   # rank 0's batch, then rank 1's
            net(mx.nd.array(x, ctx=gpu))
            torch.cuda.synchronize()
            rows.append(net._fq_stat_matrix[:, :len(x)].cpu().numpy().copy())
        cur = np.asarray([O.batch_mean(np.concatenate([q[l] for q in rows])) for l in range(layers)], np.float32)
        state = O.ema_update(state, cur, 0.9)
    fqdist.detach_calibration_sync(net)
    np.testing.assert_array_equal(r["thr"], state)


def test_cli_kl_calibration_two_ranks_equals_one_device(gpu, tmp_path):
    """KL: ranges from global batch 0 (broadcast), exact histograms summed in one all-reduce, the search on every rank:
    thresholds and the offline evaluation equal one device walking the same batches in the same order."""
    import gpu_cli_worker as W
    r = _two_ranks(tmp_path, "kl")
    cli = W.cli_module()
    thr, acc, avg, _ = W.run_flow(cli, "kl", gpu, 0, 1, W.local_batches("calib"), W.local_batches("eval"), W.LOCAL_BS)
    assert np.all(thr > 0)
    np.testing.assert_array_equal(r["thr"], thr)
    assert float(r["acc"]) == acc and float(r["avg"]) == avg


@pytest.mark.parametrize("extra", [(), ("--no-fuse",)], ids=["fused", "no-fuse"])
def test_cli_evaluation_with_batches_in_flight_equals_one_at_a_time(gpu, extra):
    """`--eval-streams 3` (the default: three evaluation batches in flight, one HIP stream each, freezing forward first and
    alone) against `--eval-streams 1`, each launched eagerly (`--eval-graph 0`) and - the default - replayed from one hipGraph
    per lane - over static input / label buffers the batches are copied into, or reading resident batches in place - (the
    first batch of a lane and the ragged last batch stay eager): same
    thresholds, same accuracies, and - since the nets are random - the same LOGITS on a batch evaluated afterwards, for the
    fused net and for the plain converted one (which never replays)."""
    import gpu_cli_worker as W
    from quantization.mxnet_amd import mx
    cli = W.cli_module()
    # nine batches: every lane several times, the last one ragged (2 samples)
    evalb = W.local_batches("eval") + W.local_batches("calib")[:5] + W.local_batches("calib")[6:]
    assert len(evalb) == 9 and len(evalb[-1][0]) == 2
    res = []
    for streams, graph, resident in ((1, 0, False), (3, 0, False), (1, 1, False), (3, 1, False), (3, 1, True)):
        # (resident: the loader's batches keep their device addresses - one graph per (lane, batch) reads them in place;
        # otherwise every batch is copied into the lane's static buffer)
        thr, acc, avg, net = W.run_flow(cli, "naive_step", gpu, 0, 1, W.local_batches("calib"), evalb, W.LOCAL_BS,
                                        extra=("--eval-streams", str(streams), "--eval-graph", str(graph)) + tuple(extra),
                                        resident=resident)
        # nine batches; eager: batch 0 (main thread), the first batch of every lane's own thread, the ragged last one
        want_replays = 0 if (extra or not graph) else (7 if streams == 1 else 5)     # (each lane launches its first batch eagerly, the ragged last one too)
        assert cli.evaluate.last_replayed == want_replays, (streams, graph, cli.evaluate.last_replayed)
        x = mx.nd.array(evalb[0][0], ctx=gpu)
        res.append((thr, acc, avg, net(x).asnumpy()))
    for r in res[1:]:
        np.testing.assert_array_equal(res[0][0], r[0])
        assert res[0][1] == r[1] and res[0][2] == r[2]
        np.testing.assert_array_equal(res[0][3], r[3])
