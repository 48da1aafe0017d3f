"""Worker for tests/test_gpu_rccl.py: two ranks, BOTH on GPU 0 (the pool's boxes have one device), torch.distributed with the
"nccl" backend = RCCL.  Runs both calibration collective modes of dist.py through the real HIP kernels and writes what each
rank ended with.      nccl_worker.py <out_dir> <local_bs>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, local_bs = sys.argv[1], int(sys.argv[2])
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    from quantization.mxnet_amd import mx, dist as fqdist
    if os.environ.get("FQ_DIST_BACKEND") == "fqcomm":
        # the library's own communicator (fq_comm_*): what a host without torch.distributed binds; no process group exists
        os.environ["FQ_DIST_FORCE_GROUP"] = "1"
        fqdist.init()
        assert not dist.is_initialized() and fqdist.group_is_live()
    else:
        dist.init_process_group("nccl", device_id=dev)
    import dist_worker as W
    ctx = mx.gpu(0)
    res = {}
    for mode in ("strict", "step"):
        net = W.make_net()
        net.collect_params().reset_ctx(ctx)
        fqdist.attach_calibration_sync(net, local_bs, strict=mode == "strict")
        net.quantize_input(enable=True, online=True)
        blocks = net.collect_quantized_blocks()
        ema = []
        for shards in W.calib_steps(mode, local_bs, world):
            net(mx.nd.array(shards[rank], ctx=ctx))
            net.update_ema()
            ema.append([float(b.input_max.data().asscalar()) for b in blocks])
        res[mode] = np.asarray(ema, np.float32)
    counters = torch.tensor([float(rank + 1), 10.0], device=dev)
    fqdist.allreduce_eval_counters(counters)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), counters=counters.cpu().numpy(), **res)
    fqdist.shutdown()


if __name__ == "__main__":
    main()
