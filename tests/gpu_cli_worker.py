"""Worker for tests/test_gpu_multirank.py: the CLI's calibration flows (examples/simulate_quantization.py: `Simulation`)
on explicit, seeded batch lists, through the REAL kernels.  Launched by torch.distributed.run with two ranks that share
GPU 0 over gloo (FQ_DIST_BACKEND=gloo FQ_DIST_SHARE_GPU=1 - the pool's boxes have one device and RCCL refuses two ranks on
it), and imported by the test itself for the one-process runs the ranks are compared with.

    gpu_cli_worker.py <out_dir> <flow>        flow: naive_strict | naive_step | kl
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LOCAL_BS = 4
SIDE = 224
MODEL = "mobilenet1.0"


def cli_module():
    spec = importlib.util.spec_from_file_location("fq_cli", os.path.join(ROOT, "examples", "simulate_quantization.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def local_batches(kind):
    """The LOCAL batches in global order (rank r of W takes r, r + W, ...).  Calibration: seven of them, the last one
    ragged - so with two ranks the last step has a 2-sample batch on rank 0 and NO batch on rank 1.  Evaluation: three."""
    rng = np.random.default_rng({"calib": 11, "eval": 5, "eval_lanes": 6}[kind])
    # ("eval_lanes": eleven batches, the last one ragged - with two ranks and three lanes each, rank 0 replays graphs on all three
    # lanes and ends on a ragged eager batch, rank 1 has one batch less)
    sizes = {"calib": [LOCAL_BS] * 6 + [2], "eval": [LOCAL_BS] * 3, "eval_lanes": [LOCAL_BS] * 10 + [3]}[kind]
    out = []
    for i, b in enumerate(sizes):
        x = (rng.standard_normal((b, 3, SIDE, SIDE)) * (1.0 + 0.2 * i)).astype(np.float32)
        y = rng.integers(0, 1000, b).astype(np.float32)
        out.append((x, y))
    return out


def global_batches(kind, world):
    """What ONE device sees when the ranks' local batches of a step are its batch (rank-major)."""
    loc = local_batches(kind)
    return [(np.concatenate([x for x, _ in loc[s:s + world]]), np.concatenate([y for _, y in loc[s:s + world]]))
            for s in range(0, len(loc), world)]


class ListLoader(object):
    def __init__(self, batches, ctx, rank=0, world=1):
        from quantization.mxnet_amd import mx
        self.total_batches = len(batches)
        self._mine = [(mx.nd.array(x, ctx=ctx), mx.nd.array(y, ctx=ctx)) for x, y in batches[rank::world]]

    def __len__(self):
        return len(self._mine)

    def __iter__(self):
        return iter(self._mine)


def options(cli, flow, extra=()):
    argv = ["--model", MODEL, "--use-gpu", "0", "--batch-size", str(LOCAL_BS), "--quantize-input-offline",
            "--calib-epoch", "1", "--pretrained", "false"] + list(extra)
    if flow == "kl":
        argv += ["--calib-mode", "kl"]
    if flow == "naive_strict":
        argv += ["--strict-global-batch"]
    return cli.parse_args(argv)


def run_flow(cli, flow, ctx, rank, world, calib, evalb, batch_size, extra=(), resident=False):
    """build -> convert -> calibrate -> freeze -> offline evaluation, exactly `Simulation.execute` with the two loaders
    replaced.  Returns (thresholds of every quantised block, acc, avg_acc)."""
    opt = options(cli, flow, extra)
    opt.batch_size = batch_size
    sim = cli.Simulation(opt, ctx, rank, world)
    np.random.seed(opt.fixed_random_seed)
    sim.build_net()
    sim.quantise_net()
    sim.train_loader = ListLoader(calib, ctx, rank, world)
    sim.eval_loader = ListLoader(evalb, ctx, rank, world)
    sim.eval_loader.resident_batches = bool(resident)     # device batches at stable addresses: graphs read them in place
    if flow == "kl":
        sim.calibrate_kl()
    else:
        sim.calibrate_naive()
    acc, avg = sim.final_evaluation(online=False)
    thr = np.asarray([float(b.input_max.data().asscalar()) for b in sim.net.collect_quantized_blocks()], np.float32)
    return thr, acc, avg, sim.net


def main():
    out_dir, flow = sys.argv[1], sys.argv[2]
    cli = cli_module()
    from quantization.mxnet_amd import dist as fqdist
    from quantization.mxnet_amd.mx import gpu
    rank, local, world = fqdist.init()
    ctx = gpu(local)
    if flow == "all_one_rank":
        # FQ_DIST_FORCE_GROUP=1, WORLD_SIZE=1: every flow of the CLI with its collectives really issued on a ONE-rank group
        # (backend "nccl" = RCCL), evaluation with three batches in flight replayed from hipGraphs
        assert world == 1 and fqdist.group_is_live()
        import torch.distributed as dist
        res = {"backend": np.asarray([ord(c) for c in dist.get_backend()])}
        try:
            evalb = local_batches("eval") + local_batches("calib")[:5] + local_batches("calib")[6:]
            for f in ("naive_step", "naive_strict", "kl"):
                thr, acc, avg, _ = run_flow(cli, f, ctx, 0, 1, local_batches("calib"), evalb, LOCAL_BS,
                                            extra=("--eval-streams", "3", "--eval-graph", "1"))
                res[f + "_thr"], res[f + "_acc"], res[f + "_avg"] = thr, np.float64(acc), np.float64(avg)
                res[f + "_replayed"] = np.int64(cli.evaluate.last_replayed)
            np.savez(os.path.join(out_dir, "one_rank.npz"), **res)
        finally:
            fqdist.shutdown()
        return
    try:
        if flow == "strict_lanes":
            # evaluation with three batches in flight per rank, replayed from hipGraphs, ragged last batch on rank 0
            thr, acc, avg, _ = run_flow(cli, "naive_strict", ctx, rank, world, local_batches("calib"),
                                        local_batches("eval_lanes"), LOCAL_BS,
                                        extra=("--eval-streams", "3", "--eval-graph", "1"))
            np.savez(os.path.join(out_dir, "rank%d.npz" % rank), thr=thr, acc=np.float64(acc), avg=np.float64(avg),
                     world=world, device=torch.cuda.current_device(), replayed=np.int64(cli.evaluate.last_replayed))
            return
        thr, acc, avg, _ = run_flow(cli, flow, ctx, rank, world, local_batches("calib"), local_batches("eval"), LOCAL_BS)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), thr=thr, acc=np.float64(acc), avg=np.float64(avg),
                 world=world, device=torch.cuda.current_device())
    finally:
        fqdist.shutdown()


if __name__ == "__main__":
    main()
