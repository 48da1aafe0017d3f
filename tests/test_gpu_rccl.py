"""The RCCL code path of dist.py executed on real hardware: two ranks that SHARE GPU 0 (the pool's boxes have a single
device) over torch.distributed's "nccl" backend.  RCCL may refuse two ranks on one device ("Duplicate GPU detected"); the
outcome is recorded either way in gpurun_out/rccl_smoke.txt, and the test is skipped — not passed — when it refuses."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_device_ema(gpu, local_bs, world, case="strict"):
    import dist_worker as W
    from quantization.mxnet_amd import mx
    net = W.make_net()
    net.collect_params().reset_ctx(gpu)
    net.quantize_input(enable=True, online=True)
    blocks = net.collect_quantized_blocks()
    ema = []
    for shards in W.calib_steps(case, local_bs, world):
        net(mx.nd.array(np.concatenate(shards), ctx=gpu))
        net.update_ema()
        ema.append([float(b.input_max.data().asscalar()) for b in blocks])
    return np.asarray(ema, np.float32)


def test_collectives_execute_on_rccl_with_one_rank(gpu, tmp_path):
    """World size 1 over the "nccl" backend: the all-gather (strict mode), the fp64 all-reduce (one collective per step) and
    the counter all-reduce really go through RCCL on the device; with one rank both modes must equal the plain run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_worker.py"), str(tmp_path), "6"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    r = np.load(os.path.join(tmp_path, "rank0.npz"))
    want = _single_device_ema(gpu, 6, 1)
    np.testing.assert_array_equal(r["strict"], want)
    np.testing.assert_array_equal(r["step"], want)
    np.testing.assert_array_equal(r["counters"], [1.0, 10.0])


def test_collectives_execute_on_the_librarys_own_communicator_with_one_rank(gpu, tmp_path):
    """FQ_DIST_BACKEND=fqcomm: dist.py over fq_comm_* (the C ABI's RCCL communicator, what a host without torch.distributed
    binds - INTEGRATION.md) instead of torch.distributed: the strict mode's all-gather (a zero-padded sum), the fp64 all-reduce
    of a calibration step and the counter all-reduce go through it on the device; with one rank both modes equal the plain run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", FQ_DIST_BACKEND="fqcomm",
               FQ_COMM_ID_FILE=os.path.join(str(tmp_path), "uid"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_worker.py"), str(tmp_path), "6"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    r = np.load(os.path.join(tmp_path, "rank0.npz"))
    want = _single_device_ema(gpu, 6, 1)
    np.testing.assert_array_equal(r["strict"], want)
    np.testing.assert_array_equal(r["step"], want)
    np.testing.assert_array_equal(r["counters"], [1.0, 10.0])
    assert not os.path.exists(os.path.join(str(tmp_path), "uid"))        # rank 0 removes the id file after the first collective


def test_two_ranks_on_one_gpu_over_rccl(gpu, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker as W
    from quantization.mxnet_amd import mx
    local_bs, world = 3, 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "nccl_worker.py"), str(tmp_path), str(local_bs)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    log = os.path.join(ROOT, "gpurun_out", "rccl_smoke.txt")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "w") as f:
        f.write("command: %s\nreturn code: %d\n--- stderr (tail) ---\n%s\n" % (" ".join(cmd), res.returncode,
                                                                             res.stderr[-4000:]))
    if res.returncode != 0:
        refused = any(k in res.stderr for k in ("Duplicate GPU", "duplicate GPU", "invalid usage", "ncclInvalidUsage"))
        if refused:
            pytest.skip("RCCL refuses two ranks on one device on this box (see gpurun_out/rccl_smoke.txt)")
        raise AssertionError(res.stderr[-3000:])
    r = [np.load(os.path.join(tmp_path, "rank%d.npz" % i)) for i in range(world)]
    for k in ("strict", "step", "counters"):
        np.testing.assert_array_equal(r[0][k], r[1][k], "ranks disagree on " + k)
    np.testing.assert_array_equal(r[0]["counters"], [3.0, 20.0])
    # strict mode == ONE device on the global batch, through the same kernels
    net = W.make_net()
    net.collect_params().reset_ctx(gpu)
    net.quantize_input(enable=True, online=True)
    blocks = net.collect_quantized_blocks()
    ema = []
    for shards in W.calib_steps("strict", local_bs, world):
        net(mx.nd.array(np.concatenate(shards), ctx=gpu))
        net.update_ema()
        ema.append([float(b.input_max.data().asscalar()) for b in blocks])
    np.testing.assert_array_equal(r[0]["strict"], np.asarray(ema, np.float32))
    np.testing.assert_array_equal(r[0]["step"][:, 0], np.asarray(ema, np.float32)[:, 0])
    with open(log, "a") as f:
        f.write("RCCL path executed: strict and one-collective-per-step modes agree across ranks; strict equals one device\n")


def test_c_abi_collectives_over_rccl_with_one_rank(gpu):
    """fq_comm_unique_id / fq_comm_init / fq_allreduce_{f32,f64,i64} / fq_comm_destroy (the collectives an
    integrator without torch.distributed binds) really go through librccl on the device; with one rank an all-reduce is
    the identity, and the calibration step built from them - fq_stat_rows_sum -> all-reduce -> fq_mean_from_sums ->
    fq_ema_update - equals the single-device batch mean + EMA.  Runs in a child process (a communicator per process)."""
    code = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
from quantization.mxnet_amd import ops
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
assert ops.comm_world() == 0
uid = ops.comm_unique_id()
assert len(uid) == 128
ops.comm_init(0, 1, uid)
assert ops.comm_world() == 1
a = torch.arange(1000, dtype=torch.float32, device=dev) * 0.5
b = torch.arange(54, dtype=torch.float64, device=dev) + 0.25
c = torch.arange(53 * 2048, dtype=torch.int64, device=dev)
for t in (a, b, c):
    want = t.clone()
    ops.comm_allreduce(t, ops.COMM_SUM)
    ops.comm_allreduce(t, ops.COMM_MAX)
    torch.cuda.synchronize()
    assert torch.equal(t, want), t.dtype
# one calibration step through the C ABI only
stats = torch.rand(5, 8, device=dev)
rec = torch.zeros(6, dtype=torch.float64, device=dev)
ops.stat_rows_sum(stats, 8, out=rec)
ops.comm_allreduce(rec)
means = torch.zeros(5, device=dev)
ops.mean_from_sums(rec, out=means)
want = torch.stack([ops.batch_mean(stats[i].contiguous())[0] for i in range(5)])
assert torch.equal(means, want)
try:
    ops.comm_init(0, 1, uid)
    raise SystemExit("second fq_comm_init did not fail")
except Exception as e:
    assert "already exists" in str(e), e
ops.comm_destroy()
assert ops.comm_world() == 0
ops.comm_destroy()
print("C-ABI collectives ok")
""" % ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "C-ABI collectives ok" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


def test_c_abi_collectives_validate_arguments():
    from quantization.mxnet_amd import _lib
    import ctypes
    assert _lib.LIB.fq_comm_world() == 0
    rc = _lib.LIB.fq_allreduce_f32(ctypes.c_void_p(8), 4, 0, None)
    assert rc != 0 and b"fq_comm_init first" in _lib.LIB.fq_last_error()
    rc = _lib.LIB.fq_comm_init(3, 2, ctypes.create_string_buffer(128))
    assert rc != 0 and b"rank 3 of 2" in _lib.LIB.fq_last_error()


def test_every_cli_flow_on_a_one_rank_rccl_group(gpu, tmp_path):
    """not only raw collectives but the CLI's FULL flows - naive-EMA calibration with the single
    all-reduce per step, the strict per-layer all-gathers, the KL range broadcast + histogram all-reduce, the counter
    all-reduce - on a process group whose backend is "nccl" (RCCL) with ONE rank, followed by the offline evaluation with
    three batches in flight replayed from hipGraphs (collectives and captured graphs on one device: the transport refuses a
    collective inside a capture, dist.py).  Thresholds and accuracies equal the same flows without any process group."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_cli_worker as W
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               FQ_DIST_FORCE_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FQ_DIST_BACKEND", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_cli_worker.py"), str(tmp_path), "all_one_rank"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    r = np.load(os.path.join(tmp_path, "one_rank.npz"))
    assert "".join(chr(c) for c in r["backend"]) == "nccl"
    cli = W.cli_module()
    evalb = W.local_batches("eval") + W.local_batches("calib")[:5] + W.local_batches("calib")[6:]
    for flow in ("naive_step", "naive_strict", "kl"):
        thr, acc, avg, _ = W.run_flow(cli, flow, gpu, 0, 1, W.local_batches("calib"), evalb, W.LOCAL_BS,
                                      extra=("--eval-streams", "1", "--eval-graph", "0"))
        np.testing.assert_array_equal(r[flow + "_thr"], thr, flow)
        assert float(r[flow + "_acc"]) == acc and float(r[flow + "_avg"]) == avg, flow
        assert int(r[flow + "_replayed"]) == 5, flow


def test_a_collective_inside_a_graph_capture_is_refused(gpu):
    """dist.py: collectives share the compute stream with forwards an evaluation loop may be capturing; one that lands in a
    capture is refused with an error that says so (instead of being replayed with every batch)."""
    import torch
    from quantization.mxnet_amd import dist as fqdist
    t = torch.zeros(4, device=gpu.torch_device)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(gpu.torch_device)
    with pytest.raises(RuntimeError, match="capturing a graph"):
        with torch.cuda.graph(g, stream=s):
            fqdist._not_while_capturing(t)
