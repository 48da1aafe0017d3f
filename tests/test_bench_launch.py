"""bench.py's own launcher (CPU part): `python bench.py --gpus N` with no WORLD_SIZE must start N ranks itself — decided
before anything touches a GPU — and hand back the children's status.  Without a GPU the ranks can only refuse, which is
exactly what shows that they were started and that their status comes back."""
import os
import subprocess
import sys

import torch

from conftest import ROOT


def _run(args, **env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_n_without_launcher_starts_ranks_and_returns_their_status():
    if torch.cuda.is_available():
        return                                    # the GPU-side test (tests/test_gpu_multirank.py) covers the real run
    res = _run(["--gpus", "2", "--steps", "1", "--no-cpu-baseline"], FQ_BENCH_SHARE_GPU="1", FQ_BENCH_BACKEND="gloo")
    assert res.returncode != 0
    # a rank ran and said so (torch.distributed.run ends the other rank as soon as the first one fails: on a loaded machine the
    # second may not get to print), and the launcher's failure report names the rank it came from
    assert res.stderr.count("bench.py needs an MI355X") >= 1, res.stderr[-2000:]
    assert "ChildFailedError" in res.stderr or "local_rank" in res.stderr, res.stderr[-2000:]


def test_gpus_n_beyond_the_nodes_devices_is_refused_by_the_parent():
    want = torch.cuda.device_count() + 1
    if want < 2:
        want = 2
    res = _run(["--gpus", str(want), "--steps", "1"])
    assert res.returncode != 0 and "GPU(s)" in res.stderr and "torch.distributed" not in res.stderr


def test_mismatched_world_is_an_error():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"],
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True,
                         timeout=300)
    assert res.returncode != 0 and "launcher started 2" in res.stderr


def test_help_texts_format():
    """argparse expands help strings with %: a bare per-cent sign in one of them only shows when --help is asked for."""
    for script in ("bench.py", os.path.join("examples", "simulate_quantization.py")):
        res = subprocess.run([sys.executable, os.path.join(ROOT, script), "--help"], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0 and "usage" in res.stdout, res.stderr[-1500:]
